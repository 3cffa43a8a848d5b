/*
 * har_scalar.cpp -- BASELINE config 1 / SURVEY.md 8 row a20: the `scalar_rgb` plumbing path (CPU, no GPU).
 *
 * SamplingIntegrator::render's NON-JIT branch (src/render/integrator.cpp:190-274): block size against the thread
 * count, Spiral::next_block (src/render/spiral.cpp:27-73), render_block (integrator.cpp:398-446: Morton pixel order,
 * per-pixel reseed, `seed *= prod(film_size)`, `seed += block_id * block_size^2`), render_sample (:448-520) and
 * ImageBlock::put's scalar branch on a bordered block with the DISCRETISED reconstruction filter
 * (src/render/imageblock.cpp:228-375, include/mitsuba/core/rfilter.h:70-79, src/core/rfilter.cpp:11-26), then
 * put_block into the film (imageblock.cpp:160-186).
 *
 * This is an explicit, separately named entry point (`har_render_scalar`, Python variant 'scalar_rgb'): the
 * `hip_ad_rgb` entry points never call it and never fall back to it.  It runs the SAME path code as the HIP
 * kernels -- the HAR_HD headers compiled for the host: BVH8 traversal, compute_si, shade_lane<MODE_PATH> -- with
 * the scalar variants' draw semantics (path.cpp:226-227 `break`, :244-249 conditional emitter samples).
 */
#include "../../include/hip_ad_rgb.h"
#include "har_cpu.h"
#include "har_path.h"
#include "har_scene_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <exception>
#include <thread>
#include <vector>

extern int har_set_error(const std::string &msg);

namespace {

using namespace har;

struct ScalarStack {                  /* traversal stack of accel_trace (depth-first bound checked by the caller) */
    static constexpr int Capacity = 64;
    uint32_t x[Capacity], y[Capacity];
    void push(int l, uint32_t a, uint32_t b) { x[l] = a; y[l] = b; }
    void pop(int l, uint32_t &a, uint32_t &b) { a = x[l]; b = y[l]; }
};

/* host view of a lowered scene: DScene over the HostScene's own vectors */
struct BoundScene {
    HostScene hs; std::vector<DTexture> dtex; DScene ds{};
    void bind() {
        dtex.clear();
        for (size_t k = 0; k < hs.textures.size(); ++k) dtex.push_back(hs.device_texture(k, hs.textures[k].data.data()));
        DScene &S = ds;
        S.accel.nodes = hs.nodes.data(); S.accel.tris = hs.tris.data(); S.accel.insts = hs.inst_recs.data(); S.accel.mesh_info = nullptr;
        S.accel.root = hs.root; S.accel.has_tlas = hs.has_tlas; S.accel.n_tris = (uint32_t) hs.tris.size(); S.accel.n_insts = (uint32_t) hs.inst_recs.size();
        S.accel.top_root = hs.top_root; S.accel.top_first = hs.top_first; S.accel.top_count = hs.top_count; S.accel.top_last = hs.top_last;
        S.blas_tri_ranges = hs.blas_tri_ranges.data();
#if HAR_SHADING_TRIS
        S.shade_tris = hs.shade_tris.data();
#endif
        S.verts = hs.verts.data(); S.faces = hs.faces.data(); S.meshes = hs.meshes.data(); S.bsdfs = hs.bsdfs.data();
        S.textures = dtex.data(); S.emitters = hs.emitters.data(); S.insts = hs.insts.data(); S.bsdf_tables = hs.bsdf_tables.data();
        S.n_emitters = (uint32_t) hs.emitters.size(); S.n_meshes = (uint32_t) hs.meshes.size();
        S.n_bsdfs = (uint32_t) hs.bsdfs.size(); S.n_insts = (uint32_t) hs.insts.size(); S.n_textures = (uint32_t) hs.textures.size();
        S.env_emitter = hs.env_emitter;
        S.bsdf_types = 0; for (const DBsdf &b : hs.bsdfs) S.bsdf_types |= (1u << b.type) | ((b.flags & BF_TWOSIDED) ? 0x80000000u : 0u);
        S.envmap = nullptr; S.emitter_cdf = hs.emitter_cdf.data();
        hs.bind_tables(S, hs.emitter_distr.data());
        if (hs.has_mesh_emitters || hs.has_point_emitters || !hs.emitter_distr.empty()) S.bsdf_types |= HAR_SCENE_ENVMAP;
        if (hs.has_envmap) { hs.envmap.tex = hs.env_tex.data(); hs.envmap.warp = hs.env_warp.data(); S.envmap = &hs.envmap; S.bsdf_types |= HAR_SCENE_ENVMAP; }
    }
};

/* dr::morton_decode<Point2u>: even bits -> x, odd bits -> y */
inline uint32_t compact_bits(uint32_t v) {
    v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu; v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0x0000ffffu;
    return v;
}

struct Block { int32_t off_x, off_y; uint32_t size_x, size_y, id; };

/* Spiral (src/render/spiral.cpp): blocks of one pass in the order next_block() hands them out */
std::vector<Block> spiral(uint32_t w, uint32_t h, uint32_t off_x, uint32_t off_y, uint32_t bs) {
    const int32_t nbx = (int32_t) ((w + bs - 1) / bs), nby = (int32_t) ((h + bs - 1) / bs);
    const uint32_t count = (uint32_t) (nbx * nby);
    std::vector<Block> out; out.reserve(count);
    int32_t px = nbx / 2, py = nby / 2; int dir = 0 /* right, down, left, up */; uint32_t steps_left = 1, spiral_size = 1;
    for (uint32_t k = 0; k < count; ++k) {
        Block b; b.id = k;
        const uint32_t ox = (uint32_t) px * bs, oy = (uint32_t) py * bs;
        b.size_x = std::min(bs, w - ox); b.size_y = std::min(bs, h - oy); b.off_x = (int32_t) (ox + off_x); b.off_y = (int32_t) (oy + off_y);
        out.push_back(b);
        if (k + 1 == count) break;
        do {
            if (dir == 0) ++px; else if (dir == 1) ++py; else if (dir == 2) --px; else --py;
            if (--steps_left == 0) {
                dir = (dir + 1) % 4;
                if (dir == 0 || dir == 2) ++spiral_size;
                steps_left = spiral_size;
            }
        } while (px < 0 || py < 0 || px >= nbx || py >= nby);
    }
    return out;
}

/* PathIntegrator::sample, scalar variant, on the host-compiled path code: radiance + valid_ray */
Vec3 sample_scalar(const DScene &S, const ShadeParams &P, uint64_t &rng, Vec3 o, Vec3 d, float maxt, bool &valid_ray, int &status) {
    Vec3 result(0.f);
    valid_ray = S.env_emitter >= 0;                               /* path.cpp:114-115 (hide_emitters is refused by the caller) */
    if (P.max_depth == 0) return result;                          /* path.cpp:102-103 */
    PathState st; st.o = o; st.d = d; st.maxt = maxt; st.throughput = Vec3(1.f); st.rng = rng; st.lane = 0; st.prev_p = Vec3(0.f); st.prev_bsdf_pdf = 1.f;
    st.flags = 1u << 16; st.eta = 1.f;
    for (;;) {
        Hit hit; ScalarStack stack;
        accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status);
        ShadeResult R;
        R.next.rng = st.rng;                                      /* a path that ends before this iteration's draws (path.cpp:226-227 `break`) leaves the stream where it is */
        shade_lane<MODE_PATH, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT>(S, P, st, hit, R);
        valid_ray = valid_ray || hit.t != HAR_INF;                /* path.cpp:307-308 */
        if (R.add_emission) result = Vec3(fma_(R.em_a.x, R.em_b.x, result.x), fma_(R.em_a.y, R.em_b.y, result.y), fma_(R.em_a.z, R.em_b.z, result.z));
        if (R.item && R.item_ray) {                                /* the emitter sample's shadow ray (path.cpp:271-281; ray_test inside sample_emitter_direction) */
            Hit h2; ScalarStack s2;
            if (!accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, h2, s2, status)) result = result + R.contrib;
        }
        if (!R.alive) { rng = R.next.rng; break; }
        st = R.next;
    }
    return valid_ray ? result : Vec3(0.f);                        /* path.cpp:341-345 */
}

} // namespace

extern "C" int har_render_scalar(const HarSceneDesc *desc, const HarSensor *sensor, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                                 uint32_t block_size, uint32_t n_threads, float *film, uint32_t *block_size_used) {
    try {
        if (!desc || !sensor || !film) return har_set_error("null argument");
        if (spp == 0) return har_set_error("spp must be > 0");
        if (max_depth < 0 && max_depth != -1) return har_set_error("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
        if (rr_depth <= 0) return har_set_error("\"rr_depth\" must be set to a value greater than zero!");
        BoundScene B; std::string e;
        if (!lower_scene(*desc, B.hs, e)) return har_set_error(e);
        if (B.hs.stack_need() > (uint32_t) ScalarStack::Capacity) return har_set_error("scene too deep for the scalar traversal stack");
        B.bind();
        DSensor C;
        if (!lower_sensor(*sensor, C, e)) return har_set_error(e);
        const DScene &S = B.ds;
        const uint32_t W = C.crop_w, H = C.crop_h;
        /* Film::sample_border: the spiral runs over the enlarged film, every block is shifted back by the border (integrator.cpp:162-165, 248-249) */
        const uint32_t Wg = C.samp_w, Hg = C.samp_h; const int32_t sample_shift = (int32_t) C.border;
        if (n_threads == 0) n_threads = har_usable_cores();         /* affinity mask and container quota, not the machine's CPU count */

        /* ReconstructionFilter::init_discretization (rfilter.cpp:11-26) */
        constexpr int RES = 31;                                   /* MI_FILTER_RESOLUTION */
        const bool box = C.rfilter == 0;
        const float radius = box ? .5f : C.radius;
        float values[RES + 1];
        for (int i = 0; i < RES; ++i) { const float x = (radius * (float) i) / (float) RES; values[i] = box ? (x <= .5f ? 1.f : 0.f) : rfilter_eval(C, x); }
        values[RES] = 0.f;
        const float scale_factor = (float) RES / radius;
        const int border = (int) std::ceil(radius - .5f - 2.f * HAR_RAY_EPS);
        auto eval_discretized = [&](float x) { const uint32_t idx = std::min<uint32_t>((uint32_t) std::fabs(x * scale_factor), (uint32_t) RES); return values[idx]; };

        /* integrator.cpp:203-214: a block for every thread */
        if (block_size == 0) {
            block_size = 32;                                      /* MI_BLOCK_SIZE */
            while (!(block_size == 1 || ((Wg + block_size - 1) / block_size) * ((Hg + block_size - 1) / block_size) >= n_threads)) block_size /= 2;
        }
        if (block_size_used) *block_size_used = block_size;
        std::vector<Block> blocks = spiral(Wg, Hg, C.crop_x, C.crop_y, block_size);
        for (Block &b : blocks) { b.off_x -= sample_shift; b.off_y -= sample_shift; }
        const uint32_t seed_scaled = seed * (Wg * Hg);              /* integrator.cpp:231: seed *= prod(film_size), film_size = crop_size (:162) */
        const ShadeParams P0{ 0u, (uint32_t) max_depth, (uint32_t) rr_depth, HAR_SHADE_SCALAR_DRAWS };

        std::mutex film_mutex; std::atomic<uint32_t> next(0); std::atomic<int> status_all(0);
        std::exception_ptr worker_error; std::mutex error_mutex;
        auto worker_body = [&]() {
            std::vector<float> blk; ShadeParams P = P0;            /* per worker: P.seed names the current pixel's stream */
            for (;;) {
                if (status_all.load()) break;                     /* an overflow reported by any worker ends the render early */
                const uint32_t bi = next.fetch_add(1);
                if (bi >= blocks.size()) break;
                const Block &b = blocks[bi];
                /* ImageBlock(size, border = true): (size + 2 * border)^2 x {R, G, B, W}, cleared (integrator.cpp:419) */
                const uint32_t bw = b.size_x + 2u * (uint32_t) border, bh = b.size_y + 2u * (uint32_t) border;
                blk.assign((size_t) bw * bh * 4, 0.f);
                const uint32_t bseed = seed_scaled + b.id * block_size * block_size;          /* integrator.cpp:412 */
                int status = 0;
                for (uint32_t i = 0; i < block_size * block_size; ++i) {
                    /* sampler->seed(seed + i) with a wavefront of one lane: Sampler::seed (sampler.cpp:129-148), lane index 0 */
                    uint64_t rng, inc; sampler_seed(bseed + i, 0u, rng, inc);
                    const uint32_t px = compact_bits(i), py = compact_bits(i >> 1);
                    if (px >= b.size_x || py >= b.size_y) continue;
                    const float pos_x = (float) ((int32_t) px + b.off_x), pos_y = (float) ((int32_t) py + b.off_y);
                    for (uint32_t j = 0; j < spp; ++j) {
                        /* render_sample (integrator.cpp:448-520) */
                        const float jx = pcg32_next_float(rng, inc), jy = pcg32_next_float(rng, inc);
                        const float spx = pos_x + jx, spy = pos_y + jy;
                        const float sx = 1.f / (float) W, sy = 1.f / (float) H;
                        Vec3 o, d; float maxt;
                        sensor_sample_ray(C, fma_(spx, sx, -(float) C.crop_x * sx), fma_(spy, sy, -(float) C.crop_y * sy), o, d, maxt);
                        P.seed = bseed + i;                       /* the stream's increment is a function of (seed value, lane 0) */
                        bool valid = false;
                        const Vec3 rgb = sample_scalar(S, P, rng, o, d, maxt, valid, status);
                        const float v[4] = { rgb.x, rgb.y, rgb.z, 1.f };
                        /* block->put(box_filter ? pos : sample_pos, aovs) */
                        const float ppx = box ? pos_x : spx, ppy = box ? pos_y : spy;
                        if (box) {                                /* imageblock.cpp:228-250 (no filter object) -- the film hands box-filtered blocks no rfilter */
                            const int32_t x = (int32_t) std::floor(ppx) - b.off_x + border, y = (int32_t) std::floor(ppy) - b.off_y + border;
                            if ((uint32_t) x < bw && (uint32_t) y < bh) { float *p = blk.data() + 4 * ((size_t) y * bw + x); for (int k = 0; k < 4; ++k) p[k] += v[k]; }
                            continue;
                        }
                        const float pfx = ppx + ((float) border - (float) b.off_x - .5f), pfy = ppy + ((float) border - (float) b.off_y - .5f);
                        const int32_t x0 = std::max((int32_t) std::ceil(pfx - radius), 0), y0 = std::max((int32_t) std::ceil(pfy - radius), 0);
                        const int32_t x1 = std::min((int32_t) std::floor(pfx + radius), (int32_t) bw - 1), y1 = std::min((int32_t) std::floor(pfy + radius), (int32_t) bh - 1);
                        if (x0 > x1 || y0 > y1) continue;
                        float wx[32], wy[32];
                        const int cx = x1 - x0 + 1, cy = y1 - y0 + 1;
                        if (cx > 32 || cy > 32) { status = HAR_STACK_OVERFLOW; continue; }
                        float relx = (float) x0 - pfx, rely = (float) y0 - pfy;
                        for (int k = 0; k < cx; ++k) { wx[k] = eval_discretized(relx); relx += 1.f; }
                        for (int k = 0; k < cy; ++k) { wy[k] = eval_discretized(rely); rely += 1.f; }
                        for (int yy = 0; yy < cy; ++yy)
                            for (int xx = 0; xx < cx; ++xx) {
                                const float w = wx[xx] * wy[yy];
                                float *p = blk.data() + 4 * ((size_t) (y0 + yy) * bw + (x0 + xx));
                                for (int k = 0; k < 4; ++k) p[k] = fma_(v[k], w, p[k]);
                            }
                    }
                }
                if (status) status_all.store(status);
                /* film->put_block (imageblock.cpp:160-186): the bordered block is added where it overlaps the crop window */
                std::lock_guard<std::mutex> lock(film_mutex);
                for (uint32_t y = 0; y < bh; ++y) {
                    const int32_t fy = b.off_y - border + (int32_t) y - (int32_t) C.crop_y;
                    if ((uint32_t) fy >= H) continue;
                    for (uint32_t x = 0; x < bw; ++x) {
                        const int32_t fx = b.off_x - border + (int32_t) x - (int32_t) C.crop_x;
                        if ((uint32_t) fx >= W) continue;
                        const float *src = blk.data() + 4 * ((size_t) y * bw + x); float *dst = film + 4 * ((size_t) fy * W + (size_t) fx);
                        for (int k = 0; k < 4; ++k) dst[k] += src[k];
                    }
                }
            }
        };
        /* an exception in a pool thread (std::bad_alloc of a block buffer) must not reach std::terminate: it is kept, the other workers stop at
         * their next block, and the calling thread rethrows it after the join (into the catch clauses below) */
        auto worker = [&]() {
            try { worker_body(); }
            catch (...) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!worker_error) worker_error = std::current_exception();
                next.store((uint32_t) blocks.size());
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; ++t) { try { pool.emplace_back(worker); } catch (...) { break; } }
        worker();
        for (auto &t : pool) t.join();
        if (worker_error) std::rethrow_exception(worker_error);
        if (status_all.load()) return har_set_error("scalar render: traversal stack / filter footprint overflow");
        return 0;
    } catch (const std::bad_alloc &) { return har_set_error("har_render_scalar: out of memory"); }
    catch (const std::exception &ex) { return har_set_error(std::string("har_render_scalar: ") + ex.what()); }
}
