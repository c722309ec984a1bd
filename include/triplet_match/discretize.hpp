/*
 * triplet_match/discretize.hpp — quantisers and the key hash of the drop-in API.
 * Mirrors include/discretize (:8-17) and include/impl/discretize.hpp (:5-56): discretize() x2 and
 * std::hash<key> = MurmurHash3_x86_32 (seed 42), here over std::array<uint32_t, Dim>.
 */
#ifndef TRIPLET_MATCH_DISCRETIZE_HPP_
#define TRIPLET_MATCH_DISCRETIZE_HPP_

#include "common.hpp"

namespace triplet_match {

struct discretization_params { /* include/discretize:8-12 */
    float distance_step_count = 20.f;
    float angle_step = 0.17453292f;
    float curvature_ratio_step_count = 0.f; /* dead in the reference */
};

inline uint32_t discretize(float value, float min_value, float range_value, uint32_t steps) {
    return tm_discretize_range(value, min_value, range_value, steps);
}
inline uint32_t discretize(float value, float step_size) { return tm_discretize_step(value, step_size); }

template <int Dim>
using key_t = std::array<uint32_t, Dim>;

namespace detail {
template <int Dim>
inline uint32_t murmur(const key_t<Dim>& key) { return tm_hash_key(key.data(), Dim); }
}  // namespace detail

}  // namespace triplet_match

namespace std {
template <size_t Dim>
struct hash<std::array<uint32_t, Dim>> {
    size_t operator()(const std::array<uint32_t, Dim>& k) const { return tm_hash_key(k.data(), (int)Dim); }
};
}  // namespace std

#endif
