"""The boundary held to the reference's own header TEXT (build container only: skipped where /root/reference is
absent, i.e. on the GPU box).

tests/test_abi.py checks that the ten names are exported and that the record sizes are the surveyor's figures.
This test reads the reference's headers themselves:

* solr/engines/cuda/CudaRayTracer.h:25-67 - the ten `extern "C"` prototypes (USE_MANAGED_MEMORY is #undef'd in
  Consts.h:23, USE_KINECT is not built): return type, name, and the type of every parameter in order must be what
  include/solr_hip.h part 1 declares;
* solr/types.h:39-329, the `#else` (CUDA) branch of `#ifdef USE_OPENCL`: typedefs, enums and every struct of the
  device contract.  Field order, field types and the offsets COMPUTED from the text with the CUDA vector_types.h
  layout rules (float2/int2 aligned 8, float3/int3 aligned 4, float4/int4 aligned 16, __align__(16) records) must
  equal what gcc reports - offsetof / sizeof, field by field, by the reference's field names - for
  include/solr_types.h;
* solr/Consts.h: every limit and named constant include/solr_types.h re-declares has the reference's value.

Nothing of the reference is copied: the test parses the files where they lie.
"""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "solr")),
                                reason="the reference tree is only present in the build container")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _resolve_conditionals(text, defined=()):
    """the text a preprocessor would keep: #ifdef / #ifndef / #else / #endif resolved with `defined` as the only
    macros that are set (neither header uses #if or #elif around what is parsed here)"""
    keep, stack = [], []
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|else|endif|if|elif)\b\s*(\w*)", line)
        if m:
            kind, macro = m.group(1), m.group(2)
            assert kind not in ("if", "elif"), line
            if kind == "ifdef":
                stack.append(macro in defined)
            elif kind == "ifndef":
                stack.append(macro not in defined)
            elif kind == "else":
                stack[-1] = not stack[-1]
            else:
                stack.pop()
            continue
        if all(stack):
            keep.append(line)
    assert not stack
    return "\n".join(keep)


def _prototypes(text, need_extern):
    """{name: (return type, [parameter types])} of the function prototypes in text"""
    out = {}
    pattern = r'%s([A-Za-z_]\w*(?:\s*\*)?)\s+(\w+)\s*\(([^;{]*?)\)\s*;' % (r'extern\s+"C"\s+' if need_extern else r'^\s*')
    for m in re.finditer(pattern, text, flags=re.S | re.M):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        types = []
        for p in [q.strip() for q in params.split(",") if q.strip()]:
            if p == "void":
                continue
            pm = re.match(r"^(.*?)(\w+)$", p, flags=re.S)          # the last word is the parameter's name
            types.append(re.sub(r"\s+", "", pm.group(1)))           # 'BoundingBox*', 'vec2i', ...
        out[name] = (re.sub(r"\s+", "", ret), types)
    return out


def test_the_ten_prototypes_are_the_reference_headers():
    ref = open(os.path.join(REF, "solr/engines/cuda/CudaRayTracer.h")).read()
    consts = open(os.path.join(REF, "solr/Consts.h")).read()
    assert re.search(r"#undef\s+USE_MANAGED_MEMORY", consts), "Consts.h:23: the managed-memory variants are compiled out"
    ref = _resolve_conditionals(_strip_comments(ref))      # neither USE_MANAGED_MEMORY nor USE_KINECT
    theirs = _prototypes(ref, need_extern=True)
    assert sorted(theirs) == sorted(["initialize_scene", "finalize_scene", "reshape_scene", "h2d_scene",
                                     "h2d_materials", "h2d_randoms", "h2d_textures", "h2d_lightInformation",
                                     "d2h_bitmap", "cudaRender"]), sorted(theirs)
    mine = open(os.path.join(ROOT, "include/solr_hip.h")).read()
    part1 = mine[mine.index("Part 1 - the reference boundary"):mine.index("Part 2 - extensions")]
    ours = _prototypes(_strip_comments(part1), need_extern=False)
    assert sorted(ours) == sorted(theirs), "part 1 of include/solr_hip.h is exactly the reference's boundary"
    for name, (ret, types) in theirs.items():
        assert ours[name] == (ret, types), (name, ours[name], (ret, types))
    # by value: SceneInfo, PostProcessingInfo and the vectors are not pointers on either side
    assert theirs["cudaRender"][1] == ["vec2i", "vec4i", "SceneInfo", "vec4i", "PostProcessingInfo", "vec3f", "vec3f",
                                       "vec4f"]
    # ... and the whole header is inside one extern "C" block
    assert re.search(r'extern\s+"C"\s*\{', mine)


# ---- solr/types.h ---------------------------------------------------------------------------------------------
# CUDA's vector_types.h (the header the reference's #else branch includes): size, alignment
CUDA_VECTORS = {"float": (4, 4), "int": (4, 4), "float2": (8, 8), "float3": (12, 4), "float4": (16, 16),
                "int2": (8, 8), "int3": (12, 4), "int4": (16, 16), "unsigned char": (1, 1)}


def _cuda_branch(text):
    """solr/types.h as the CUDA build sees it: the `#else` branch of `#ifdef USE_OPENCL` (types.h:29-76)"""
    assert re.search(r"#ifdef\s+USE_OPENCL\b.*?cl_float4.*?\n#else\b.*?vector_types\.h.*?\n#endif", text, flags=re.S), \
        "types.h:29-76 no longer has the two flavours"
    return _resolve_conditionals(_strip_comments(text), defined=("USE_CUDA",))


def _parse_types(text):
    """(typedefs {alias: base}, enums {name}, structs {name: (aligned16, [(type, field, count)])})"""
    text = _strip_comments(text)
    typedefs = {a: b.strip() for b, a in re.findall(r"typedef\s+([\w ]+?)\s+(\w+)\s*;", text)}
    enums = set(re.findall(r"\benum\s+(\w+)\s*\{", text))
    structs = {}
    for m in re.finditer(r"\bstruct\s+(__ALIGN16__\s+)?(\w+)\s*\{(.*?)\}\s*;", text, flags=re.S):
        fields = []
        for decl in [d.strip() for d in m.group(3).split(";") if d.strip()]:
            fm = re.match(r"^([\w ]+?)\s*(\*?)\s*(\w+)\s*(?:\[(\d+)\])?$", decl, flags=re.S)
            assert fm, decl
            fields.append((fm.group(1).strip() + fm.group(2), fm.group(3), int(fm.group(4) or 1)))
        structs[m.group(2)] = (bool(m.group(1)), fields)
    return typedefs, enums, structs


def _layout(structs, typedefs, enums, name):
    """[(field, offset, size)], size, alignment of a struct by the C layout rules over CUDA_VECTORS"""
    def size_align(t):
        if t.endswith("*"):
            return 8, 8
        seen = 0
        while t in typedefs and t not in CUDA_VECTORS:
            t = typedefs[t]
            seen += 1
            assert seen < 8
        if t in enums:
            return 4, 4
        if t in CUDA_VECTORS:
            return CUDA_VECTORS[t]
        if t in structs:
            _, s, a = _layout(structs, typedefs, enums, t)
            return s, a
        raise AssertionError("type %r of the reference's header is not one the layout rules know" % t)

    aligned16, fields = structs[name]
    offset, align, rows = 0, (16 if aligned16 else 1), []
    for t, field, count in fields:
        s, a = size_align(t)
        offset = (offset + a - 1) // a * a
        rows.append((field, offset, s * count))
        offset += s * count
        align = max(align, a)
    return rows, (offset + align - 1) // align * align, align


RECORDS = ["SceneInfo", "LightInformation", "Material", "BoundingBox", "Primitive", "TextureInfo",
           "PostProcessingInfo", "PostProcessingBuffer"]


def test_record_layouts_are_the_reference_headers():
    text = _cuda_branch(open(os.path.join(REF, "solr/types.h")).read())
    assert re.search(r"#define\s+__ALIGN16__\s+__align__\(16\)", text), "types.h:75"
    typedefs, enums, structs = _parse_types(text)
    assert typedefs["vec3f"] == "float3" and typedefs["vec4f"] == "float4" and typedefs["vec2i"] == "int2"
    assert typedefs["PrimitiveXYIdBuffer"] == "int4" and typedefs["BitmapBuffer"] == "unsigned char"
    assert typedefs["Lamp"] == "int" and typedefs["RandomBuffer"] == "float"
    for r in RECORDS:
        assert r in structs, r

    # what gcc says about include/solr_types.h, asked field by field with the REFERENCE's field names
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "solr_types.h"', 'int main(void) {']
    for r in RECORDS:
        lines.append('printf("%s %%zu %%zu\\n", sizeof(%s), _Alignof(%s));' % (r, r, r))
        for _, field, _ in structs[r][1]:
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));'
                         % (r, field, r, field, r, field))
    for v in ("vec2f", "vec3f", "vec4f", "vec2i", "vec3i", "vec4i", "PrimitiveXYIdBuffer", "BitmapBuffer", "Lamp",
              "RandomBuffer"):
        lines.append('printf("%s %%zu %%zu\\n", sizeof(%s), _Alignof(%s));' % (v, v, v))
    lines += ["return 0; }"]
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "layout.c"), os.path.join(tmp, "layout")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    measured = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in out.splitlines()}

    for r in RECORDS:
        rows, size, align = _layout(structs, typedefs, enums, r)
        assert measured[r] == (size, align), (r, measured[r], (size, align))
        for field, offset, fsize in rows:
            assert measured["%s.%s" % (r, field)] == (offset, fsize), (r, field, measured["%s.%s" % (r, field)],
                                                                          (offset, fsize))
        # and include/solr_types.h declares no field the reference does not have, in the same order
        mine = _strip_comments(open(os.path.join(ROOT, "include/solr_types.h")).read())
        body = re.search(r"typedef\s+struct[^{]*?%s_s\s*\{(.*?)\}\s*%s\s*;" % (r, r), mine, flags=re.S).group(1)
        names = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            decl = re.sub(r"\[\d+\]", "", decl)
            first = re.match(r"^[\w ]+?\s*\*?\s*(\w+)\s*(,.*)?$", decl, flags=re.S)
            names.append(first.group(1))
            if first.group(2):
                names += [n.strip() for n in first.group(2).split(",") if n.strip()]
        assert names == [f for _, f, _ in structs[r][1]], (r, names)
    for v in ("vec2f", "vec3f", "vec4f", "vec2i", "vec3i", "vec4i"):
        assert measured[v] == CUDA_VECTORS[typedefs[v]], v
    assert measured["PrimitiveXYIdBuffer"] == (16, 16) and measured["BitmapBuffer"] == (1, 1)
    assert measured["Lamp"] == (4, 4) and measured["RandomBuffer"] == (4, 4)
    # the figures SURVEY.md appendix C quotes follow from the text as well
    sizes = {r: _layout(structs, typedefs, enums, r)[1] for r in RECORDS}
    assert sizes == {"SceneInfo": 112, "LightInformation": 48, "Material": 176, "BoundingBox": 48, "Primitive": 128,
                     "TextureInfo": 32, "PostProcessingInfo": 16, "PostProcessingBuffer": 32}, sizes


def test_enumerations_are_the_reference_headers():
    def enumerators(text):
        out = {}
        for m in re.finditer(r"\benum\s+(\w+)\s*\{(.*?)\}", _strip_comments(text), flags=re.S):
            value, items = -1, {}
            for item in [i.strip() for i in m.group(2).split(",") if i.strip()]:
                if "=" in item:
                    n, v = [x.strip() for x in item.split("=")]
                    value = int(v)
                else:
                    n, value = item, value + 1
                items[n] = value
            out[m.group(1)] = items
        return out
    theirs = enumerators(_cuda_branch(open(os.path.join(REF, "solr/types.h")).read()))
    ours = enumerators(open(os.path.join(ROOT, "include/solr_types.h")).read())
    for name in ("CameraType", "FrameBufferType", "AdvancedIllumination", "GraphicsLevel", "AtmosphericEffect",
                 "PrimitiveType", "TextureType", "PostProcessingType"):
        assert ours[name] == theirs[name], name


def test_limits_and_named_constants_are_the_reference_headers():
    consts = _strip_comments(open(os.path.join(REF, "solr/Consts.h")).read())
    theirs = {}
    for n, v in re.findall(r"#define\s+(\w+)\s+([^\n]+)", consts):
        theirs[n] = v.strip()
    for n, v in re.findall(r"const\s+(?:unsigned\s+)?int\s+(\w+)\s*=\s*([^;]+);", consts):
        theirs[n] = v.strip()
    mine = dict((n, v.strip()) for n, v in
                re.findall(r"#define\s+(\w+)\s+([^\n]+)", _strip_comments(open(os.path.join(ROOT, "include/solr_types.h")).read())))

    def value(table, name, depth=0):
        expr = table[name]
        assert depth < 8
        expr = re.sub(r"[A-Za-z_]\w*", lambda m: str(value(table, m.group(0), depth + 1)) if m.group(0) in table
                      else m.group(0), expr)
        expr = re.sub(r"(\d)f\b", r"\1", expr)
        return eval(expr, {"__builtins__": {}})          # integer / float arithmetic of the two headers only

    same_name = ["NB_MAX_ITERATIONS", "BOUNDING_BOXES_TREE_DEPTH", "NB_MAX_BOXES", "NB_MAX_PRIMITIVES", "NB_MAX_LAMPS",
                 "NB_MAX_MATERIALS", "NB_MAX_TEXTURES", "NB_MAX_FRAMES", "NB_MAX_LIGHTINFORMATIONS", "MAX_BITMAP_WIDTH",
                 "MAX_BITMAP_HEIGHT", "MAX_BITMAP_SIZE", "MATERIAL_NONE", "TEXTURE_NONE", "TEXTURE_MANDELBROT",
                 "TEXTURE_JULIA", "STANDARD_LUNINANCE_STRENGTH", "SKYBOX_LUNINANCE_STRENGTH", "RANDOM_MATERIALS_OFFSET",
                 "DEFAULT_LIGHT_MATERIAL", "WHITE_MATERIAL", "RED_MATERIAL", "GREEN_MATERIAL", "BLUE_MATERIAL"]
    for n in same_name:
        assert value(mine, n) == value(theirs, n), (n, mine[n], theirs[n])
    for ours_name, theirs_name in (("SOLR_MAX_GPU_COUNT", "MAX_GPU_COUNT"), ("SOLR_COLOR_DEPTH", "gColorDepth"),
                                   ("SOLR_PI", "PI")):
        assert value(mine, ours_name) == value(theirs, theirs_name), ours_name
