/* tm_table.h — device-side probe of the model's key index (open addressing on
 * MurmurHash3_x86_32(key, seed 42), the reference's std::hash for the key type:
 * include/impl/discretize.hpp:5-56). Shared by tm_model.hip and tm_scene.hip. */
#ifndef TM_TABLE_H_
#define TM_TABLE_H_

#include "tm_math.h"

namespace tmx {

__device__ __forceinline__ uint32_t ht_find(const uint32_t* __restrict__ ht_slot,
                                            const uint4* __restrict__ ht_key, uint32_t cap_mask,
                                            key4 k) {
    uint32_t h = murmur_key4(k.k0, k.k1, k.k2, k.k3) & cap_mask;
    for (;;) {
        uint32_t s = ht_slot[h];
        if (s == 0u) return 0xFFFFFFFFu;
        uint4 q = ht_key[h];
        if (q.x == k.k0 && q.y == k.k1 && q.z == k.k2 && q.w == k.k3) return s - 1u;
        h = (h + 1u) & cap_mask;
    }
}


}  // namespace tmx

#endif /* TM_TABLE_H_ */
