/*
 * triplet_match/projected_search.hpp — the traits search driver: model and scene looked at through a
 * projector (plane2_traits, cylinder_traits, ...), as the reference's former model<Proj,Point> /
 * scene<Proj,Point> design intended (remnants: include/impl/model.hpp:194-201,283-305, include/scene:40-45;
 * the traits themselves: include/impl/{plane2,cylinder}_traits.hpp). In this snapshot of the reference the
 * traits are compiled but not wired, so the wiring below is ours; what each step
 * computes is the reference's:
 *
 *   model side   the model cloud is projected once by its own projector (Traits::project / tangent / normal)
 *                and a model<Point> is built on the uvw cloud; correspondences use the traits'
 *                intrinsic_distance (tm_model_set_metric: plane2 = 2-D, cylinder = wrap-around in u).
 *   scene side   Traits::sample_count scene points are drawn (seeded mt19937, from the points no earlier
 *                projector has explained) -> Traits::init_from_samples (nullptr = rejected, e.g. plane2's
 *                normal-consistency test, include/impl/plane2_traits.hpp:63-68) -> the resident scene is
 *                mapped into that projector's uvw space on the device (tm_scene_create_projected: one pass
 *                over HBM) -> the identity search (search::find_all: sampling, descriptor match, scoring,
 *                ICP, arbitration) runs on the uvw scene against the uvw model -> the matches' scene points
 *                are masked for the projectors that follow.
 *
 * Cheap rejections before a projector gets a search (ours): a cylinder whose radius differs from the
 * model's by more than radius_tolerance; a projector under which fewer scene points project than
 * model_match_factor x |model| (it could never collect enough correspondences); a projector whose surface
 * holds fewer tangent points than the arbitration demands as tangent inliers (min_points, :114,227).
 * The first sample is drawn anywhere, the others inside its neighbourhood of one model diameter.
 */
#ifndef TRIPLET_MATCH_PROJECTED_SEARCH_HPP_
#define TRIPLET_MATCH_PROJECTED_SEARCH_HPP_

#include <random>

#include "model.hpp"
#include "search.hpp"
#include "traits.hpp"

namespace triplet_match {

/* how scene<Point>::find_all_parallel<Traits> looks for projectors */
struct projector_search {
    uint32_t max_draws = 20000;      /* sample sets drawn at most (most fall to the cheap rejections) */
    uint32_t max_searches = 32;      /* projectors that get a full search at most */
    uint32_t max_idle_searches = 8;  /* stop after this many searched projectors in a row without a match */
    uint32_t max_idle_draws = 250;   /* ... or after this many sample sets in a row that led to no new search (rejected,
                                        or on a surface that has had its search) */
    float radius_tolerance = 0.05f;  /* cylinder: |r_scene - r_model| <= tolerance * r_model */
    bool on_surface_samples = true;  /* sample points of the search: |w| <= the model projector's threshold */
    bool skip_searched_surfaces = true; /* a sample set lying on a surface that already had its search is passed over:
                                           that search ran its rounds until nothing more was found there (find_all,
                                           include/impl/scene.hpp:70-79), and a coincident projector sees the same
                                           points. "Coincident": all samples within the model projector's threshold of
                                           the earlier surface and the surfaces' axes within ~3 degrees. */
    uint64_t seed = 13;
    bool verbose = false;
};

/* model<Point> in its projector's uvw space */
template <typename Traits, typename Point>
class projected_model {
public:
    typedef pointcloud<Point> cloud_t;
    typedef typename Traits::const_handle_t const_handle_t;

    /* `handle` = the model's projector (Traits::init_from_model / init_from_samples / init_from_axis).
     * Points whose project() is nullopt are dropped; original_index() maps back. */
    projected_model(typename cloud_t::ConstPtr cloud, const_handle_t handle, discretization_params params,
                    int device = 0)
        : projected_model(cloud, handle, params, detail::default_context(device)) {}
    /* on a context of the caller's (make_context): the scene it is searched in lives on the same one */
    projected_model(typename cloud_t::ConstPtr cloud, const_handle_t handle, discretization_params params,
                    context_t ctx)
        : cloud_(cloud), handle_(handle) {
        static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
        if (!ctx) throw std::invalid_argument("projected_model: null context");
        if (!cloud_ || cloud_->size() == 0 || !handle_) throw std::runtime_error("projected_model: empty cloud or projector");
        const uint32_t n = (uint32_t)cloud_->size();
        float st[TM_STATE_FLOATS] = {0};
        handle_->pack(st);
        std::vector<float> uvw(12 * (size_t)n);
        std::vector<int32_t> valid(n);
        detail::check(tm_traits_project_cloud(ctx->ctx, Traits::kind, st, cloud_->points[0].raw(), n, uvw.data(),
                                              valid.data()),
                      "tm_traits_project_cloud");
        uvw_cloud_ = cloud_t::empty();
        for (uint32_t i = 0; i < n; ++i) {
            if (!valid[i]) continue;
            Point p;
            memcpy(p.raw(), uvw.data() + 12 * (size_t)i, 48);
            uvw_cloud_->points.push_back(p);
            kept_.push_back(i);
        }
        if (uvw_cloud_->size() < 2) throw std::runtime_error("projected_model: the projector rejects the model cloud");
        model_.reset(new model<Point>(uvw_cloud_, params, ctx));
        model_->disable_curvature_filter(); /* the tangents were mapped into uvw space: |tangent| > 0.7 only */
        detail::check(tm_model_set_metric(model_->handle(), Traits::kind, handle_->radius), "tm_model_set_metric");
    }
    void init(const sample_parameters& sp) { model_->init(sp); }
    model<Point>& uvw_model() { return *model_; }
    const const_handle_t& handle() const { return handle_; }
    typename cloud_t::ConstPtr cloud() const { return cloud_; }
    typename cloud_t::ConstPtr uvw_cloud() const { return uvw_cloud_; }
    uint32_t original_index(uint32_t uvw_index) const { return kept_[uvw_index]; }
    float valid_fraction() const { return (float)kept_.size() / (float)cloud_->size(); }

private:
    typename cloud_t::ConstPtr cloud_;
    const_handle_t handle_;
    typename cloud_t::Ptr uvw_cloud_;
    std::vector<uint32_t> kept_;
    std::unique_ptr<model<Point>> model_;
};

/* Draws Traits::sample_count distinct points (seeded) until init_from_samples accepts them and the cloud
 * projects almost completely under the result: the projector of a model that lies on one surface of the
 * traits' kind. `threshold` is what the reference keeps in the model state (cylinder: 5 x resolution,
 * include/impl/cylinder_traits.hpp:19; plane: 2 x max |w|, include/impl/plane_traits.hpp:34-39). */
template <typename Traits, typename Point>
inline typename Traits::handle_t fit_model_projector(const pointcloud<Point>& cloud, float threshold, uint64_t seed = 13,
                                                     float min_valid_fraction = 0.99f, uint32_t max_draws = 1000) {
    auto mh = std::make_shared<typename Traits::state_t>();
    mh->threshold = threshold;
    std::mt19937 rng((uint32_t)seed);
    const uint32_t n = (uint32_t)cloud.size();
    if (n < (uint32_t)Traits::sample_count) return nullptr;
    for (uint32_t draw = 0; draw < max_draws; ++draw) {
        uint32_t idx[Traits::sample_count];
        for (int k = 0; k < Traits::sample_count; ++k) idx[k] = rng() % n;
        bool distinct = true;
        for (int a = 0; a < Traits::sample_count; ++a)
            for (int b = a + 1; b < Traits::sample_count; ++b) distinct = distinct && idx[a] != idx[b];
        if (!distinct) continue;
        float flat[12 * Traits::sample_count], ms[TM_STATE_FLOATS] = {0}, st[TM_STATE_FLOATS] = {0};
        for (int k = 0; k < Traits::sample_count; ++k) memcpy(flat + 12 * k, cloud.points[idx[k]].raw(), 48);
        mh->pack(ms);
        if (tm_traits_init_from_samples(Traits::kind, ms, flat, st)) continue;
        uint32_t ok = 0, tried = 0;
        for (uint32_t i = 0; i < n; i += std::max(1u, n / 2000u), ++tried) { /* a sample of the cloud is enough */
            float uvw[3];
            const float xyz[3] = {cloud.points[i].x, cloud.points[i].y, cloud.points[i].z};
            ok += tm_traits_project(Traits::kind, st, xyz, uvw) ? 1u : 0u;
        }
        if ((float)ok < min_valid_fraction * (float)tried) continue;
        auto h = std::make_shared<typename Traits::state_t>();
        h->unpack(st);
        return h;
    }
    return nullptr;
}

}  // namespace triplet_match

#endif
