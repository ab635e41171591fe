/*
 * SWCReader.h - neuron morphologies in SWC format (reference: solr/io/SWCReader.{h,cpp}).
 *
 * An SWC file lists sample points "id type x y z radius parent".  The reference turns them into spheres
 * and cylinders: the root (parent -1) becomes a sphere of 1.5 x its radius; every other sample whose
 * parent is not the root adds a cylinder from the sample to its parent (with the SAMPLE's radius) and a
 * sphere at the parent (with the parent's radius); samples hanging directly off the root add nothing.
 * Its way of reading is kept as it is (SWCReader.cpp:55-83): one line is dropped, then seven blank-
 * separated words are taken - across line ends if need be - and the rest of the line they end on is
 * dropped with the next round; a record whose first word is "#" is a comment.  Header lines therefore
 * pair up or split depending on their word counts, and data lines (exactly seven words) read one per
 * line once the header is through.  Coordinates: (float)(scale * (position + value)) in binary64,
 * radius (float)(scale.w * value).
 */
#pragma once

#include <map>
#include <string>

#include "GPUKernel.h"

namespace solr
{
struct Morphology
{
    int branch = 0;
    float x = 0.f, y = 0.f, z = 0.f, radius = 0.f;
    int parent = 0;
    int primitiveId = 0;
};
typedef std::map<int, Morphology> Morphologies;

class SWCReader
{
public:
    CPUBoundingBox loadMorphologyFromFile(const std::string &filename, GPUKernel &kernel, const vec4f &position,
                                          const vec4f &scale, const int materialId);
    Morphologies getMorphologies() { return m_morphologies; }

private:
    Morphologies m_morphologies; /* kept across calls, like the reference's */
};
}
