/*
 * PDBReader.h - molecules from Protein Data Bank files (reference: solr/io/PDBReader.{h,cpp}).
 *
 * What the reference's reader does, restated:
 *   - materials 0..118 become the element colours of its table (flat, specular 1 / 100);
 *   - every line that starts with ATOM is an atom; the fields are cut at fixed positions and blanks inside
 *     a field are dropped: serial = characters 7..10, atom code 13..16, chain = character 21 - '@',
 *     residue 23..25, x 31..36, y 39..44, z 47..52 (negated), element 77..78 (0-based; the reference's
 *     position-driven scanner drops the characters at 6 11 12 17 21 22 26 30 37 38 45 46 53 76 79, so a
 *     coordinate keeps two of its three decimals and a two-letter element symbol its second letter - kept,
 *     it decides where the atoms are and which colour they get).  A field whose closing position lies
 *     beyond the end of the line stays 0 / empty;
 *   - element -> material id (table row, or by chain / residue for materialType 1 / 2) and radius (table,
 *     picometres; first match), the molecule's extent and centre;
 *   - geometry by GeometryType: a sphere per atom; for the stick types a half-bond cylinder from every
 *     atom to the midpoint of every other atom closer than 1.7 (3.4 along a backbone) with the same
 *     backbone flag; everything scaled by scale / extent and spread by 2 x 30 around the centre.
 * The element tables are data (sol-r_amd/host/pdb_elements.txt, tools/gen_pdb_elements.py), read once from
 * next to this library or from $SOLR_PDB_ELEMENTS.
 */
#pragma once

#include <string>

#include "GPUKernel.h"

namespace solr
{
/* reference: PDBReader.h:29-37 */
enum GeometryType
{
    gtAtoms = 0,
    gtFixedSizeAtoms = 1,
    gtSticks = 2,
    gtAtomsAndSticks = 3,
    gtIsoSurface = 4,
    gtBackbone = 5
};

class PDBReader
{
public:
    PDBReader() : m_nbPrimitives(0), m_nbBoxes(0) {}
    virtual ~PDBReader() {}

    /* returns the size of the scaled molecule; (0, 0, 0, -1) if the element tables cannot be found */
    vec4f loadAtomsFromFile(const std::string &filename, GPUKernel &kernel, GeometryType geometryType,
                            const float defaultAtomSize, const float defaultStickSize, const int materialType,
                            const vec4f scale, const bool useModels = false);
    int getNbBoxes() { return m_nbBoxes; }
    int getNbPrimitives() { return m_nbPrimitives; }

private:
    int m_nbPrimitives;
    int m_nbBoxes;
};
}
