"""The hand-scheduled node loops (sol-r_amd/csrc/rt_device.h advanceTidyShallow / advanceTidy) name fixed registers -
s[64:94], v[58:63]: sub-registers of pairs cannot be named through operands.  What keeps the compiler from holding
a live value in one of them across the statement is the clobber list alone, and a register added to the template
but not to the list would corrupt a frame silently.  This test makes that loud: the device source is run through
the preprocessor (the templates are assembled from macros), every `asm volatile` statement is taken apart, and

  * every register the template names literally must be in the statement's clobber list (or be exec / vcc / scc
    / a named special register),
  * every clobbered register must be named by the template (a stale entry costs the allocator a register),
  * vcc and scc must be declared when the template writes them,
  * the fixed registers are the documented banks and no operand is bound to a register by name.

With a complete clobber list the compiler cannot hold a value in these registers across the statement (that is the
contract of the list, whatever the allocator does in a later release); what the list cannot catch is an edit of the
template, and that is what is checked.  No GPU needed: hipcc only preprocesses here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = os.path.join(ROOT, "sol-r_amd", "csrc", "rows", "everything.hip")  # rt_device.h + renderer_kernel.h
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _preprocessed(tmp_path):
    out = str(tmp_path / "solr_hip.ii")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-E", "-std=c++17", SOURCE, "-o", out],
                   check=True, cwd=str(tmp_path))
    return open(out).read()


def _statements(text):
    """[(template, outputs, inputs, clobbers)] of every asm volatile statement"""
    found = []
    for m in re.finditer(r"asm volatile\(", text):
        i, depth, in_string, parts, start = m.end(), 1, False, [], m.end()
        while depth:
            c = text[i]
            if in_string:
                if c == "\\":
                    i += 1
                elif c == '"':
                    in_string = False
            elif c == '"':
                in_string = True
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            elif c == ":" and depth == 1:
                parts.append(text[start:i])
                start = i + 1
            i += 1
        parts.append(text[start:i - 1])
        template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0])).replace("\\n", "\n").replace("\\t", "\t")
        clobbers = re.findall(r'"([^"]+)"', parts[3]) if len(parts) > 3 else []
        found.append((template, parts[1] if len(parts) > 1 else "", parts[2] if len(parts) > 2 else "", clobbers))
    return found


def _literal_registers(template):
    """registers the template names itself (not through an operand)"""
    regs = set()
    body = re.sub(r"%\[\w+\]|%\d+|%=", " ", template)
    for kind, a, b in re.findall(r"\b([sv])\[(\d+):(\d+)\]", body):
        regs.update("%s%d" % (kind, k) for k in range(int(a), int(b) + 1))
    body = re.sub(r"\b[sv]\[\d+:\d+\]", " ", body)
    for kind, a in re.findall(r"(?<![\w.])([sv])(\d+)\b", body):
        regs.add("%s%d" % (kind, int(a)))
    return regs


@pytest.fixture(scope="module")
def statements(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    return _statements(_preprocessed(tmp_path_factory.mktemp("asm")))


def test_there_are_the_statements_we_think(statements):
    loops = [s for s in statements if "s_load_dwordx8" in s[0]]
    # the two-bank loop, and the three-bank loop in its three forms (any record / sorted bounds / sorted, read in reverse)
    assert len(loops) == 4, [s[0][:40] for s in statements]
    assert len(statements) >= 3


def test_every_fixed_register_is_clobbered_and_every_clobber_is_used(statements):
    for template, _outputs, _inputs, clobbers in statements:
        named = _literal_registers(template)
        listed = {c for c in clobbers if re.fullmatch(r"[sv]\d+", c)}
        assert named <= listed, "named by the template but not clobbered: %s" % sorted(named - listed)
        assert listed <= named, "clobbered but not named by the template: %s" % sorted(listed - named)
        if re.search(r"\bv_cmpx?_\w+_e32\b|\bvcc\b", template):
            assert "vcc" in clobbers
        if re.search(r"\bs_(cmp|add|sub|lshl|lshr|and|or|xor|cselect|min|max)\w*\b", template):
            assert "scc" in clobbers


def test_the_fixed_banks_are_the_documented_ones(statements):
    """DESIGN / the comment above the loops: s[64:94], v[58:63]"""
    allowed = {"s%d" % k for k in range(64, 95)} | {"v%d" % k for k in range(58, 64)}
    for template, _o, _i, clobbers in statements:
        assert _literal_registers(template) <= allowed
        assert {c for c in clobbers if re.fullmatch(r"[sv]\d+", c)} <= allowed


def test_operands_do_not_ask_for_fixed_registers(statements):
    """nothing binds an operand to a register by name (register asm variables would bypass the list)"""
    text = open(os.path.join(ROOT, "sol-r_amd", "csrc", "rt_device.h")).read()
    assert not re.search(r"register\s+\w+\s+\w+\s+asm\s*\(", text)


# ---- what the COMPILER makes of the statements (a ROCm upgrade must fail here, not in a frame) ------------------------
ROW = os.path.join(ROOT, "sol-r_amd", "csrc", "rows", "sphere_plane.hip")     # the Cornell box's kernels: both loop forms


def _makefile_flags():
    text = open(os.path.join(ROOT, "sol-r_amd", "Makefile")).read().replace("\\\n", " ")
    numeric = re.search(r"^NUMERIC\s*=\s*(.*)$", text, re.M).group(1).split()
    flags = re.search(r"^HIPFLAGS\s*=\s*(.*)$", text, re.M).group(1)
    flags = flags.replace("$(NUMERIC)", " ".join(numeric)).replace("$(ARCH)", "gfx950")
    return [f for f in flags.split() if f not in ("-fPIC",)]


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    """the device assembly of one row file, built with the Makefile's own flags"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "row.s")
    subprocess.run([HIPCC] + _makefile_flags() + ["--cuda-device-only", "-S", "-o", out, ROW], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text):
    """{mangled name: body} of the k_standardRenderer instantiations"""
    out = {}
    for m in re.finditer(r"^(_Z18k_standardRenderer\w+):\s*;", text, re.M):
        end = text.index(".Lfunc_end", m.end())
        out[m.group(1)] = text[m.end():end]
    return out


def _loop_bodies(body):
    """the instruction lines of every node-loop statement the compiler emitted into a kernel: from the `s_mov_b64 ..., exec`
    that opens a statement to its LX_ label (the labels carry the statement's unique number)"""
    loops = []
    for m in re.finditer(r"^LX_(\d+):", body, re.M):
        number = m.group(1)
        first = body.index("LWA_%s:" % number)
        opening = body.rfind("s_mov_b64 s[", 0, first)
        lines = [l.strip() for l in body[opening:m.start()].split("\n")]
        loops.append([l for l in lines if l and not l.startswith(";") and not l.startswith(".") and not l.startswith("//")])
    return loops


def test_the_compiler_emits_every_loop_whole_and_apart(statements, generated):
    """Each inlined walk has a node loop of its own, emitted verbatim: the same instruction sequence as the template (the
    operands bound), one per walk and never merged with the other walk's - whatever a later compiler's passes make of two
    identical `asm volatile` statements.  (The Makefile's -mllvm -simplifycfg-sink-common=false is not about these: that
    pass merged stores to different fields of a hit record into one store through a selected address and pinned the
    record in scratch; the scratch check below is what notices it coming back.)"""
    templates = {}
    for template, _o, _i, _c in statements:
        if "s_load_dwordx8" not in template:
            continue
        lines = [l.strip() for l in template.split("\n") if l.strip()]
        lines = lines[:lines.index("LX_%=:")]                    # (as _loop_bodies cuts the generated code)
        opcodes = [l.split()[0] for l in lines if not l.endswith(":")]
        templates[len(opcodes)] = opcodes
    assert len(templates) == 3                                   # the two-bank form, the three-bank form with and without its min / max
    kernels = _kernels(generated)
    assert len(kernels) >= 4, list(kernels)
    for name, body in kernels.items():
        loops = _loop_bodies(body)
        # a closest-hit walk and a shadow walk at least (the census kernels walk without the loop: none)
        census = "ILi1E" in name
        assert (len(loops) == 0) if census else (len(loops) >= 2), (name, len(loops))
        for loop in loops:
            opcodes = [l.split()[0] for l in loop if not l.endswith(":")]
            assert len(opcodes) in templates, (name, len(opcodes), sorted(templates))
            assert opcodes == templates[len(opcodes)], name


def test_the_fixed_registers_exist_in_every_kernel(generated):
    """s64 ... s94 and v58 ... v63 are named by number: a kernel compiled for fewer registers than that (a launch bound, an
    occupancy attribute) would let the assembler use registers the allocator never reserved"""
    meta = re.findall(r"\.name:\s+(_Z18k_standardRenderer\w+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)",
                      generated)
    assert len(meta) >= 4
    for name, sgprs, vgprs in meta:
        if "ILi1E" in name:
            continue
        assert int(sgprs) >= 95 + 6, (name, sgprs)               # s0 ... s94 and vcc / flat_scratch / xnack
        assert int(vgprs) >= 64, (name, vgprs)
    # ... and nothing of the loops went to scratch: the lean kernels hold their state in registers and LDS
    for name, spill in re.findall(r"\.name:\s+(_Z18k_standardRenderer\w+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", generated):
        if "ILi0E" in name and "ILi33E" in name:
            assert int(spill) == 0, (name, spill)
