// flood_bvh.hpp - layout of the implicit box tree shared by the tree and cell kernels (internal).
#pragma once
#include "flood_common.hpp"

namespace flooder {

constexpr int LEAF = FLOODER_BVH_LEAF;        // points per leaf
constexpr int FAN = FLOODER_BVH_FANOUT;       // children per inner node (= wave size)
constexpr int MAXL = FLOODER_BVH_MAX_LEVELS;  // levels incl. leaves

struct Levels {
  int n_levels;          // level 0 = leaves ... level n_levels-1 = top (<= 64 nodes)
  int64_t off[MAXL];     // first node of the level in the node array (levels padded to x64)
  int64_t count[MAXL];   // real nodes of the level
};

inline Levels make_levels(int64_t n_pts) {
  Levels lv;
  memset(&lv, 0, sizeof(lv));
  int64_t c = (n_pts + LEAF - 1) / LEAF;
  if (c < 1) c = 1;
  int64_t off = 0;
  int l = 0;
  for (;;) {
    lv.count[l] = c;
    lv.off[l] = off;
    off += (c + FAN - 1) / FAN * FAN;
    ++l;
    if (c <= FAN || l == MAXL) break;
    c = (c + FAN - 1) / FAN;
  }
  lv.n_levels = l;
  return lv;
}

inline int64_t total_nodes(const Levels& lv) {
  const int t = lv.n_levels - 1;
  return lv.off[t] + (lv.count[t] + FAN - 1) / FAN * FAN;
}


}  // namespace flooder
