// Device-side types of the tape — free of host headers, because this file and
// tape_interp.h are compiled twice: by hipcc into libslpx.so and at run time by hipRTC as
// part of the prelude of the generated kernel (tape_jit.cpp).
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#else
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef int int32_t;
#endif

#include "tape_ops.h"

namespace slpx {

// Leaf binding: bit 31 set -> constant pool index, else input vector index.
constexpr uint32_t kLeafConstFlag = 0x80000000u;

struct TapeTask {
  uint32_t n_leaf, n_node, n_slot;
  uint32_t leaf_off;   // into leaf_src
  uint32_t node_off;   // into node_rec (x3)
  uint32_t lvl_off;    // into lvl_ptr (n_lvl + 1 entries, local node indices)
  uint32_t n_lvl;
  uint32_t slot_off;   // into slot_edge_ptr (n_slot + 1 entries, edge indices relative to edge_off)
  uint32_t slvl_off;   // into slvl_ptr (n_slvl + 1 entries, local slot indices)
  uint32_t n_slvl;
  uint32_t edge_off;   // into edges
  uint32_t vout_off, n_vout;  // into vout_*
  uint32_t jout_off, n_jout;  // into jout_*
  uint32_t scratch_off;       // GLOBAL tasks: offset (doubles) into the scratch buffer
  uint32_t lds_doubles;       // working-set size in doubles
  uint32_t n_edge;            // number of adjoint edges
  uint32_t lds_bytes;         // LDS-staged kernel: working set + staged program
};

struct TapeEdge {
  uint32_t parent_slot;  // local slot index
  uint32_t partial;      // 2 * local interior node index + side
};

// Pointers handed to the tape kernel by value
struct TapeDev {
  const TapeTask* tasks;
  const uint32_t* leaf_src;
  const double* consts;
  const uint32_t* node_rec;
  const uint32_t* lvl_ptr;
  const uint32_t* slot_edge_ptr;
  const uint32_t* slvl_ptr;
  const TapeEdge* edges;
  const uint32_t* vout_src;
  const uint32_t* vout_dst;
  const int32_t* vout_scale;
  const uint32_t* jout_slot;
  const uint32_t* jout_dst;
  const int32_t* jout_scale;
  const uint16_t* node_rec16;
  const uint16_t* slot_edge_ptr16;
  const uint16_t* edges16;
};

}  // namespace slpx
