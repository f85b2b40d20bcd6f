// Shared plain types of the psim HIP library.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(PSIM_HOSTLOOP)
// Test-only build: kernel bodies run as host loops (tests/hostloop).  Never shipped.
#define PSIM_HD
#define PSIM_FORCEINLINE inline
#define PSIM_RESTRICT __restrict__
#include "lane_hostloop.h"
#else
#include <hip/hip_runtime.h>
#define PSIM_HD __host__ __device__
#define PSIM_FORCEINLINE __forceinline__
#define PSIM_RESTRICT __restrict__
#include "lane_hip.h"
#endif

namespace psim {

struct alignas(16) vec2 {
    double x, y;
};

// One element of the pre-order quadtree array: an internal node or a particle
// leaf.  skip = index of the first element after this element's subtree.  32 bytes,
// so a wave-uniform fetch is one 32-B load.
// tag: 0 for a leaf.  For an internal node of depth d it is the high word of the
// FP64 number the walk compares squared distances with, root_tag - (d << 21):
// halving the cell width per level divides its square by 4, i.e. subtracts 2 from
// the exponent field, which starts at bit 20 of the high word (the low word is the
// same for every depth and is kept by the walk).
struct alignas(32) NodeRec {
    double mass, x, y;
    uint32_t skip;
    uint32_t tag;
};
constexpr uint32_t kLeafTag = 0;
PSIM_HD PSIM_FORCEINLINE uint32_t node_tag(uint32_t root_tag, int depth) { return root_tag - ((uint32_t)depth << 21); }
PSIM_HD PSIM_FORCEINLINE int tag_depth(uint32_t root_tag, uint32_t tag) { return (int)((root_tag - tag) >> 21); }

// One particle in cell-sorted ("slot") order: everything a neighbour test needs
// in a single 32-byte record (position, integer radius, particle index, cell).
struct alignas(32) CellBody {
    double x, y;
    int32_t r;
    uint32_t idx;
    uint32_t cell;
    uint32_t cxy;      // cell column | cell row << 16
};

constexpr int kKeyLevels = 32;          // Morton key: 32 levels x 2 bits
constexpr int kMaxInternalDepth = 31;   // children of depth-31 nodes are bucket leaves

// ordering key of a candidate pair inside one collision sweep
// (cell | phase:4 | a | b, field widths chosen per step), see make_pair_key()
constexpr uint32_t kMaxCells = 1u << 24;
constexpr int32_t kMaxGridDim = 65535;   // cell column / row must fit 16 bits (CellBody::cxy)

enum StageBits : int32_t {
    kStageCollisions = 1,
    kStageConduction = 2,
    kStageCooling = 4,
    kStageWalls = 8,
    kStageExplosions = 16,
    kStageAll = 31,
};

// What the two gathers of a step (into Morton order for the tree, into cell order for the
// grid) need of a particle, in one 32-byte record: they read at random particle indices, and
// the memory system moves at least 64 bytes per access however little is asked for, so
// position (16 B) and mass or radius (8 / 4 B) from separate arrays cost two such accesses
// per particle and this record one.  Written by DriftKeys.
struct alignas(32) PackedBody {
    double x, y, m;
    int32_t r;
    uint32_t unused;
};

// per-contact, per-side record of the collision solver's velocity pass
struct alignas(32) VelState {
    vec2 vel;
    double temp;
    double unused;
};

}  // namespace psim
