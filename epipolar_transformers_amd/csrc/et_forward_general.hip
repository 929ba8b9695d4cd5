// libepipolar_amd.so: the operator's parameterised / pooled / prior branches as ONE kernel (et_epipolar_forward_general).
// ----------------------------------------------------------------------------
// SURVEY.md row N4.  The reference applies its optional 1x1 convolutions to the MAPS before sampling
// (epipolar.py:138-153: other1 = phi(feat2), feat1 = theta(feat1), other2 = g(feat2)), so these branches are the
// headline operator with three differences:
//   * the query rows, the similarity map and the value map are three tensors (C/BOTTLENECK and C/BOTTLENECK channels);
//   * POOLING (epipolar.py:200-202, 211-213): `view(2, K/2, C, H, W).max(0)` -- the per-channel maximum of samples k and
//     k + K/2, on both sampled maps, before the dot product: K/2 similarities / attention weights per pixel;
//   * PRIOR (epipolar.py:300-301, 308-309): a learned (K', H, W) map per camera pair added to the masked similarity, or
//     multiplied onto the soft-max output (PRIORMUL).
// The reference (and the torch restatement Epipolar._attend_general_chunk) materialises the sampled K x C x H x W tensor
// per pair; here one wave owns one reference pixel and nothing but the outputs is written:
//   lanes <-> samples : epipolar segment, the K samples' four taps and bilinear weights -> LDS (36 bytes per sample)
//   lanes <-> channels: sim[k'] = sum_c q[c] * pooled sample of the similarity map   (wave reduction per sample)
//   lanes <-> samples : `== 0 -> -1e10` mask, prior, soft-max (or / K'), first arg-max -> attn, corr_pos
//   lanes <-> channels: out[c] = sum_k' attn[k'] * pooled sample of the value map
// HBM/L2-bound gather like the per-pixel headline kernel (algorithmic bytes per pixel: K x 4 taps x (Cs + Cv) x 4 B);
// not tuned further -- these modes are not on BASELINE.json's metric.  et_epipolar_backward_general (below) is its
// backward, for every branch (round 5: cosine similarity, ATTENTION max, additive / multiplicative prior incl. the prior
// tables' own gradient, SIMILARITY prior).
#include "et_common.h"

namespace {
#include "et_wave_reduce.h"

struct GeneralParams {
    EtLayerDesc d;          // geometry, K, soft-max switches (d.C is not used)
    const float *xs, *ys, *steps, *cam;
    const float *q;         // (N, H*W, cs)   query rows: feat1 or theta(feat1), channels last
    const float *m_sim;     // (N, H*W, cs)   similarity map: feat2 or phi(feat2)
    const float *m_val;     // (N, H*W, cv)   value map: feat2 or g(feat2)
    const float *prior;     // (N, K', H*W) or NULL
    float *out;             // (N, H*W, cv)
    float *attn;            // (N, K', H*W) or NULL
    float *corr;            // (N, H*W, 2) or NULL
    int cs, cv, prior_mul;
    int cosine;             // similarity = cosine of q and the (pooled) sample (F.cosine_similarity, eps 1e-8) instead of the dot product
    int attn_max;           // ATTENTION max: raw cosine similarity, no mask / soft-max; out = the arg-max sample's values
    int sim_prior;          // SIMILARITY prior (epipolar.py:288-289): the prior table IS the attention -- no similarity, mask or soft-max
};

constexpr int kGenWaves = 4;        // waves (= reference pixels) per block
constexpr int kGenMaxQ = 8;         // query channels a lane keeps in registers: cs <= 512
constexpr int gen_wave_floats(int K) { return (K * 9 + 3) & ~3; }   // 36 bytes per sample, 16-byte aligned per wave
constexpr int gen_bwd_wave_floats(int K) { return (K * 11 + 3) & ~3; }   // + a second and a third [K'] array
constexpr size_t gen_lds_bytes(int K) { return (size_t)kGenWaves * gen_wave_floats(K) * sizeof(float); }

// one channel of one bilinear sample: tap[r] < 0 <=> outside the image (weight 0, zero padding)
__device__ __forceinline__ float gen_sample(const float *map, int ch, int c, const int4 t, const float4 w)
{
    float v = 0.f;
    if (t.x >= 0) v = fmaf(map[(size_t)t.x * ch + c], w.x, v);
    if (t.y >= 0) v = fmaf(map[(size_t)t.y * ch + c], w.y, v);
    if (t.z >= 0) v = fmaf(map[(size_t)t.z * ch + c], w.z, v);
    if (t.w >= 0) v = fmaf(map[(size_t)t.w * ch + c], w.w, v);
    return v;
}

template <bool POOL>
__global__ __launch_bounds__(kWave *kGenWaves) void epipolar_fwd_general_kernel(const GeneralParams p)
{
    extern __shared__ float s_dyn[];
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, K = d.K, HW = H * W;
    const int Ks = POOL ? K / 2 : K;          // similarities per pixel
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gp = (long long)blockIdx.x * kGenWaves + wave;
    const int n = (int)(gp / HW);
    if (n >= d.N) return;                     // wave-uniform; the kernel has no block-wide barrier
    const int pix = (int)(gp - (long long)n * HW);
    const int h = pix / W, w = pix - h * W;
    float4 *s_w = reinterpret_cast<float4 *>(s_dyn + (size_t)wave * gen_wave_floats(K));   // [K] bilinear weights
    int4 *s_tap = reinterpret_cast<int4 *>(s_w + K);                          // [K] taps
    float *s_sim = reinterpret_cast<float *>(s_tap + K);                      // [K'] similarity, then attention
    const float neg_inf = -__builtin_huge_valf();

    // ---- lanes <-> samples: the epipolar segment and every sample's taps ------------------------------------------
    const et::Segment seg = et::epipolar_segment(d, p.cam + (size_t)n * ET_CAM_STRIDE, p.xs[w], p.ys[h]);
    for (int k = lane; k < K; k += kWave) {
        const et::SampleSetup su = et::sample_setup(d, seg, p.steps[k]);
        s_w[k] = make_float4(su.weight[0], su.weight[1], su.weight[2], su.weight[3]);
        s_tap[k] = make_int4(su.tap[0], su.tap[1], su.tap[2], su.tap[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- lanes <-> channels: similarities (epipolar.py:294-295 on the pooled samples) -----------------------------
    if (!p.sim_prior) {
        const float *qrow = p.q + ((size_t)n * HW + pix) * p.cs;
        const float *m1 = p.m_sim + (size_t)n * HW * p.cs;
        float qv[kGenMaxQ];
#pragma unroll
        for (int i = 0; i < kGenMaxQ; ++i) qv[i] = (lane + i * kWave < p.cs) ? qrow[lane + i * kWave] : 0.f;
        float qq = 0.f;
        if (p.cosine) {
#pragma unroll
            for (int i = 0; i < kGenMaxQ; ++i) qq = fmaf(qv[i], qv[i], qq);
            qq = wave_all_sum(qq);
        }
        for (int k = 0; k < Ks; ++k) {
            const int4 t0 = s_tap[k];
            const float4 w0 = s_w[k];
            int4 t1 = t0;
            float4 w1 = w0;
            if (POOL) {
                t1 = s_tap[k + Ks];
                w1 = s_w[k + Ks];
            }
            float dot = 0.f, vv = 0.f;
#pragma unroll
            for (int i = 0; i < kGenMaxQ; ++i) {
                const int c = lane + i * kWave;
                if (c < p.cs) {
                    float v = gen_sample(m1, p.cs, c, t0, w0);
                    if (POOL) v = fmaxf(v, gen_sample(m1, p.cs, c, t1, w1));
                    dot = fmaf(qv[i], v, dot);
                    vv = fmaf(v, v, vv);
                }
            }
            dot = wave_all_sum(dot);
            if (p.cosine) {   // epipolar.py:282-286 / :290-293: x1 . x2 / (max(|x1|, eps) max(|x2|, eps)), eps = 1e-8
                vv = wave_all_sum(vv);
                dot = dot / (fmaxf(sqrtf(qq), 1e-8f) * fmaxf(sqrtf(vv), 1e-8f));
            }
            if (lane == 0) s_sim[k] = dot;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- lanes <-> samples: mask, prior, soft-max, arg-max (epipolar.py:298-311, :237-241) -------------------------
    {
        constexpr int KPL = 4;   // K' <= 256
        float l[KPL], a[KPL], pr[KPL];
        float vmax = neg_inf;
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            const bool in = k < Ks;
            float v = (in && !p.sim_prior) ? s_sim[k] : 0.f;
            pr[s] = (in && p.prior) ? p.prior[((size_t)n * Ks + k) * HW + pix] : 0.f;
            if (p.sim_prior) {
                v = pr[s];                                                        // :288-289: returned as is
            } else if (!p.attn_max) {                                             // (ATTENTION max: the raw cosine, :282-286)
                v = (v == 0.f) ? -1e10f : v;                                      // epipolar.py:298
                if (p.prior && !p.prior_mul) v += pr[s];                          // :300-301
                v = d.softmax_enabled ? v * d.softmax_scale : v / (float)Ks;      // :306 / :311
            }
            l[s] = v;
            if (in) vmax = fmaxf(vmax, v);
        }
        if (d.softmax_enabled && !p.attn_max && !p.sim_prior) {
            vmax = wave_all_max(vmax);
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                a[s] = (s * kWave + lane < Ks) ? expf(l[s] - vmax) : 0.f;
                sum += a[s];
            }
            sum = wave_all_sum(sum);
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                a[s] = a[s] / sum;
                if (p.prior && p.prior_mul) a[s] *= pr[s];                        // :308-309
            }
        } else {
#pragma unroll
            for (int s = 0; s < KPL; ++s) a[s] = (s * kWave + lane < Ks) ? l[s] : 0.f;
        }
        // first maximum over k' (torch.argmax): largest value, then lowest index
        float bestv = neg_inf, bestk = 1e9f;
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            if (k < Ks && a[s] > bestv) {
                bestv = a[s];
                bestk = (float)k;
            }
        }
        const float bm = wave_all_max(bestv);
        const int besti = (int)wave_all_min((bestv == bm) ? bestk : 1e9f);
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            if (k < Ks) {
                // ATTENTION max gathers the arg-max sample (epipolar.py:232-235): a one-hot weight for the sum below
                s_sim[k] = p.attn_max ? (k == besti ? 1.f : 0.f) : a[s];
                if (p.attn) p.attn[((size_t)n * Ks + k) * HW + pix] = a[s];
            }
        }
        if (p.corr && lane == 0 && besti < Ks) {
            // the location of sample `besti` of the UNPOOLED list: sample_locs[idx], idx < K' (epipolar.py:239)
            const et::SampleSetup su = et::sample_setup(d, seg, p.steps[besti]);
            float *o = p.corr + ((size_t)n * HW + pix) * 2;
            o[0] = et::de_normalize(d, su.nx, W);
            o[1] = et::de_normalize(d, su.ny, H);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- lanes <-> channels: the weighted sum of the pooled value samples (epipolar.py:243) ------------------------
    {
        const float *m2 = p.m_val + (size_t)n * HW * p.cv;
        float *orow = p.out + ((size_t)n * HW + pix) * p.cv;
        for (int c0 = 0; c0 < p.cv; c0 += 4 * kWave) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < Ks; ++k) {
                const float ak = s_sim[k];
                if (p.attn_max && ak == 0.f) continue;      // wave-uniform (the one-hot weights of ATTENTION max)
                const int4 t0 = s_tap[k];
                const float4 w0 = s_w[k];
                int4 t1 = t0;
                float4 w1 = w0;
                if (POOL) {
                    t1 = s_tap[k + Ks];
                    w1 = s_w[k + Ks];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + lane + i * kWave;
                    if (c < p.cv) {
                        float v = gen_sample(m2, p.cv, c, t0, w0);
                        if (POOL) v = fmaxf(v, gen_sample(m2, p.cv, c, t1, w1));
                        acc[i] = fmaf(ak, v, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + lane + i * kWave;
                if (c < p.cv) orow[c] = acc[i];
            }
        }
    }
}

// ---- backward of the same branches: d q, d map_sim, d map_val, d prior from d out -----------------------------------
// One wave per reference pixel again.  With P_k' / V_k' the pooled samples of the similarity / value map, g = d out:
//     s_k'  = q . P_k'            (dot)      or   q . P_k' / (max(|q|, eps) max(|P_k'|, eps))     (cosine, eps = 1e-8)
//     v_k'  = mask(s_k') [+ prior_k']                          (`== 0 -> -1e10`, epipolar.py:298, 300-301)
//     sm    = soft-max(scale v)   a = sm [* prior]             (:303-309)        or   a = v / K'   (:310-311)
//     out   = sum_k' a_k' V_k'                                 (:243)
//  =>  d a_k' = g . V_k'      d sm = d a [* prior]      d v = scale sm (d sm - sum_j sm_j d sm_j)     (or d a / K')
//      d prior_k' = d v_k' (added) | sm_k' d a_k' (multiplied)          d s_k' = d v_k', 0 under the mask
//      dot:     d q = sum d s_k' P_k'                       d P_k' = d s_k' q
//      cosine:  d q = sum alpha_k' P_k' - (sum gamma_k') q  d P_k' = alpha_k' q - beta_k' P_k'
//               alpha = d s / (|q| |P|),  beta = d s s / |P|^2,  gamma = d s s / |q|^2  (torch's cosine_similarity divides by the
//               clamped norms but differentiates the unclamped ones; a norm below eps only occurs at s = 0: masked)
//      d V_k' = a_k' g
//  ATTENTION max (:222-235, 282-286): out = V of the first arg-max of the raw cosine -- a one-hot a, no gradient through the
//  similarity.  SIMILARITY prior (:288-289): a = the prior table itself: d prior_k' = d a_k', nothing through q / P.
//  d P / d V go through the per-channel maximum to the sample that won (the first on a tie, as torch.max), times the four
//  bilinear weights onto the taps of the maps: float atomics (the sums over pixels arrive in any order: reproducible to
//  rounding only, like the tile backward).  The similarities are recomputed (nothing but the inputs is saved by the forward).
struct GeneralBwdParams {
    GeneralParams f;        // inputs as in the forward (out / attn / corr unused)
    const float *gout;      // (N, H*W, cv)
    float *gq;              // (N, H*W, cs)  written
    float *gsim;            // (N, H*W, cs)  nullable, accumulated with atomics: zero it first
    float *gval;            // (N, H*W, cv)  nullable, accumulated with atomics: zero it first
    float *gprior;          // (N, K', H*W)  nullable, written (one wave owns a pixel's K' entries)
};

__device__ __forceinline__ void gen_scatter(float *gmap, int ch, int c, const int4 t, const float4 w, float g)
{
    if (t.x >= 0) unsafeAtomicAdd(gmap + (size_t)t.x * ch + c, g * w.x);
    if (t.y >= 0) unsafeAtomicAdd(gmap + (size_t)t.y * ch + c, g * w.y);
    if (t.z >= 0) unsafeAtomicAdd(gmap + (size_t)t.z * ch + c, g * w.z);
    if (t.w >= 0) unsafeAtomicAdd(gmap + (size_t)t.w * ch + c, g * w.w);
}

template <bool POOL>
__global__ __launch_bounds__(kWave *kGenWaves) void epipolar_bwd_general_kernel(const GeneralBwdParams bp)
{
    extern __shared__ float s_dyn[];
    const GeneralParams &p = bp.f;
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, K = d.K, HW = H * W;
    const int Ks = POOL ? K / 2 : K;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gp = (long long)blockIdx.x * kGenWaves + wave;
    const int n = (int)(gp / HW);
    if (n >= d.N) return;
    const int pix = (int)(gp - (long long)n * HW);
    const int h = pix / W, w = pix - h * W;
    float4 *s_w = reinterpret_cast<float4 *>(s_dyn + (size_t)wave * gen_bwd_wave_floats(K));
    int4 *s_tap = reinterpret_cast<int4 *>(s_w + K);
    float *s_sim = reinterpret_cast<float *>(s_tap + K);   // [K'] q . P, then alpha
    float *s_a = s_sim + K;                                 // [K'] d a, then a
    float *s_b = s_a + K;                                   // [K'] |P|^2 (cosine), then beta
    const float neg_inf = -__builtin_huge_valf();
    const float kEps = 1e-8f;

    const et::Segment seg = et::epipolar_segment(d, p.cam + (size_t)n * ET_CAM_STRIDE, p.xs[w], p.ys[h]);
    for (int k = lane; k < K; k += kWave) {
        const et::SampleSetup su = et::sample_setup(d, seg, p.steps[k]);
        s_w[k] = make_float4(su.weight[0], su.weight[1], su.weight[2], su.weight[3]);
        s_tap[k] = make_int4(su.tap[0], su.tap[1], su.tap[2], su.tap[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float *qrow = p.q + ((size_t)n * HW + pix) * p.cs;
    const float *grow = bp.gout + ((size_t)n * HW + pix) * p.cv;
    const float *m1 = p.m_sim + (size_t)n * HW * p.cs;
    const float *m2 = p.m_val + (size_t)n * HW * p.cv;
    float qv[kGenMaxQ];
#pragma unroll
    for (int i = 0; i < kGenMaxQ; ++i) qv[i] = (lane + i * kWave < p.cs) ? qrow[lane + i * kWave] : 0.f;
    float qq = 0.f;
    if (p.cosine) {
#pragma unroll
        for (int i = 0; i < kGenMaxQ; ++i) qq = fmaf(qv[i], qv[i], qq);
        qq = wave_all_sum(qq);
    }

    // ---- lanes <-> channels: q . P_k' (and |P_k'|^2)  and  d a_k' = g . V_k' ------------------------------------------
    for (int k = 0; k < Ks; ++k) {
        const int4 t0 = s_tap[k];
        const float4 w0 = s_w[k];
        int4 t1 = t0;
        float4 w1 = w0;
        if (POOL) {
            t1 = s_tap[k + Ks];
            w1 = s_w[k + Ks];
        }
        float dot = 0.f, vv = 0.f, da = 0.f;
        if (!p.sim_prior) {
#pragma unroll
            for (int i = 0; i < kGenMaxQ; ++i) {
                const int c = lane + i * kWave;
                if (c < p.cs) {
                    float v = gen_sample(m1, p.cs, c, t0, w0);
                    if (POOL) v = fmaxf(v, gen_sample(m1, p.cs, c, t1, w1));
                    dot = fmaf(qv[i], v, dot);
                    vv = fmaf(v, v, vv);
                }
            }
            dot = wave_all_sum(dot);
            if (p.cosine) vv = wave_all_sum(vv);
        }
        if (!p.attn_max) {
            for (int c = lane; c < p.cv; c += kWave) {
                float v = gen_sample(m2, p.cv, c, t0, w0);
                if (POOL) v = fmaxf(v, gen_sample(m2, p.cv, c, t1, w1));
                da = fmaf(grow[c], v, da);
            }
            da = wave_all_sum(da);
        }
        if (lane == 0) {
            s_sim[k] = dot;
            s_a[k] = da;
            s_b[k] = vv;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- lanes <-> samples: a (recomputed), d prior, and the coefficients alpha / beta / gamma of the maps' gradients ----
    float gamma = 0.f;
    {
        constexpr int KPL = 4;
        const float nq = fmaxf(sqrtf(qq), kEps);
        float sim[KPL], da[KPL], pr[KPL], np_[KPL], aout[KPL], alpha[KPL], beta[KPL], dprior[KPL];
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            const bool in = k < Ks;
            const float raw = in ? s_sim[k] : 0.f;
            np_[s] = p.cosine ? fmaxf(sqrtf(in ? s_b[k] : 1.f), kEps) : 1.f;
            sim[s] = p.cosine ? raw / (nq * np_[s]) : raw;           // (the forward's expression)
            da[s] = in ? s_a[k] : 0.f;
            pr[s] = (in && p.prior) ? p.prior[((size_t)n * Ks + k) * HW + pix] : 0.f;
            if (p.sim_prior) sim[s] = pr[s];
            aout[s] = alpha[s] = beta[s] = dprior[s] = 0.f;
        }
        if (p.attn_max) {
            // first maximum over k' (torch.argmax): a one-hot weight, nothing flows through the similarity
            float bestv = neg_inf, bestk = 1e9f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const int k = s * kWave + lane;
                if (k < Ks && sim[s] > bestv) {
                    bestv = sim[s];
                    bestk = (float)k;
                }
            }
            const float bm = wave_all_max(bestv);
            const int besti = (int)wave_all_min((bestv == bm) ? bestk : 1e9f);
#pragma unroll
            for (int s = 0; s < KPL; ++s) aout[s] = (s * kWave + lane == besti) ? 1.f : 0.f;
        } else if (p.sim_prior) {
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                aout[s] = pr[s];
                dprior[s] = da[s];
            }
        } else {
            bool masked[KPL];
            float v[KPL], dv[KPL];
            const bool padd = p.prior && !p.prior_mul, pmul = p.prior && p.prior_mul;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                masked[s] = (sim[s] == 0.f);
                v[s] = masked[s] ? -1e10f : sim[s];
                if (padd) v[s] += pr[s];
            }
            if (d.softmax_enabled) {
                float l[KPL], sm[KPL], dsm[KPL];
                float vmax = neg_inf;
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    l[s] = v[s] * d.softmax_scale;
                    if (s * kWave + lane < Ks) vmax = fmaxf(vmax, l[s]);
                }
                vmax = wave_all_max(vmax);
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    sm[s] = (s * kWave + lane < Ks) ? expf(l[s] - vmax) : 0.f;
                    sum += sm[s];
                }
                sum = wave_all_sum(sum);
                float ada = 0.f;
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    sm[s] = sm[s] / sum;
                    dsm[s] = pmul ? pr[s] * da[s] : da[s];
                    ada = fmaf(sm[s], dsm[s], ada);
                }
                ada = wave_all_sum(ada);
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    dv[s] = d.softmax_scale * sm[s] * (dsm[s] - ada);
                    aout[s] = pmul ? sm[s] * pr[s] : sm[s];
                    dprior[s] = pmul ? sm[s] * da[s] : (padd ? dv[s] : 0.f);
                }
            } else {
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    const bool in = s * kWave + lane < Ks;
                    aout[s] = in ? v[s] / (float)Ks : 0.f;
                    dv[s] = da[s] / (float)Ks;
                    dprior[s] = padd ? dv[s] : 0.f;                       // (PRIORMUL only acts behind the soft-max, :308-309)
                }
            }
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const float ds = (masked[s] || s * kWave + lane >= Ks) ? 0.f : dv[s];
                if (p.cosine) {
                    alpha[s] = ds / (nq * np_[s]);
                    beta[s] = (np_[s] > kEps) ? ds * sim[s] / (np_[s] * np_[s]) : 0.f;
                    gamma += (nq > kEps) ? ds * sim[s] / (nq * nq) : 0.f;
                } else {
                    alpha[s] = ds;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            if (k < Ks) {
                s_sim[k] = alpha[s];
                s_b[k] = beta[s];
                s_a[k] = aout[s];
                if (bp.gprior) bp.gprior[((size_t)n * Ks + k) * HW + pix] = dprior[s];
            }
        }
        gamma = wave_all_sum(gamma);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- lanes <-> channels: d q, and the scatter of d P / d V through the maximum onto the maps ------------------------
    {
        float dq[kGenMaxQ];
#pragma unroll
        for (int i = 0; i < kGenMaxQ; ++i) dq[i] = 0.f;
        float *g1 = bp.gsim ? bp.gsim + (size_t)n * HW * p.cs : nullptr;
        float *g2 = bp.gval ? bp.gval + (size_t)n * HW * p.cv : nullptr;
        for (int k = 0; k < Ks; ++k) {
            const float al = s_sim[k], be = s_b[k], ak = s_a[k];
            const int4 t0 = s_tap[k];
            const float4 w0 = s_w[k];
            int4 t1 = t0;
            float4 w1 = w0;
            if (POOL) {
                t1 = s_tap[k + Ks];
                w1 = s_w[k + Ks];
            }
            if (al != 0.f || be != 0.f) {    // wave-uniform
#pragma unroll
                for (int i = 0; i < kGenMaxQ; ++i) {
                    const int c = lane + i * kWave;
                    if (c < p.cs) {
                        const float v0 = gen_sample(m1, p.cs, c, t0, w0);
                        bool second = false;
                        float v = v0;
                        if (POOL) {
                            const float v1 = gen_sample(m1, p.cs, c, t1, w1);
                            second = v1 > v0;
                            v = second ? v1 : v0;
                        }
                        dq[i] = fmaf(al, v, dq[i]);
                        if (g1) gen_scatter(g1, p.cs, c, second ? t1 : t0, second ? w1 : w0, al * qv[i] - be * v);
                    }
                }
            }
            if (g2 && ak != 0.f) {
                for (int c = lane; c < p.cv; c += kWave) {
                    bool second = false;
                    if (POOL) second = gen_sample(m2, p.cv, c, t1, w1) > gen_sample(m2, p.cv, c, t0, w0);
                    gen_scatter(g2, p.cv, c, second ? t1 : t0, second ? w1 : w0, ak * grow[c]);
                }
            }
        }
        float *gqrow = bp.gq + ((size_t)n * HW + pix) * p.cs;
#pragma unroll
        for (int i = 0; i < kGenMaxQ; ++i)
            if (lane + i * kWave < p.cs) gqrow[lane + i * kWave] = dq[i] - gamma * qv[i];
    }
}

// SIMILARITY prior: the prior table is the attention (needs one; excludes the other similarity / prior switches)
int check_sim_prior(const char *who, int flags, bool has_prior)
{
    if (!(flags & ET_GENERAL_SIM_PRIOR)) return 0;
    if (!has_prior) return fail("%s: SIM_PRIOR without a prior", who);
    if (flags & (ET_GENERAL_PRIOR_MUL | ET_GENERAL_COSINE))
        return fail("%s: SIM_PRIOR excludes PRIOR_MUL / COSINE (flags=%d)", who, flags);
    return 0;
}
}  // namespace

extern "C" {

int et_epipolar_forward_general(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                const float *cam, const float *q, const float *map_sim, const float *map_val,
                                const float *prior, int c_sim, int c_val, int flags, float *out, float *attn,
                                float *corr_pos, void *stream)
{
    if (!desc) return fail("et_epipolar_forward_general: desc is NULL");
    EtLayerDesc chk = *desc;
    chk.C = 4;                                   // (the channel counts of this entry point are c_sim / c_val)
    if (int e = validate(&chk)) return e;
    if (!xs || !ys || !steps || !cam || !q || !map_sim || !map_val || !out)
        return fail("et_epipolar_forward_general: NULL pointer");
    if (c_sim <= 0 || c_sim > kGenMaxQ * kWave) return fail("et_epipolar_forward_general: c_sim=%d outside [1, %d]", c_sim, kGenMaxQ * kWave);
    if (c_val <= 0 || c_val > 4096) return fail("et_epipolar_forward_general: c_val=%d outside [1, 4096]", c_val);
    if (flags & ~(ET_GENERAL_POOLING | ET_GENERAL_PRIOR_MUL | ET_GENERAL_COSINE | ET_GENERAL_ATTENTION_MAX | ET_GENERAL_SIM_PRIOR))
        return fail("et_epipolar_forward_general: unknown flag bits %d", flags);
    if ((flags & ET_GENERAL_ATTENTION_MAX) && prior && !(flags & ET_GENERAL_SIM_PRIOR))
        return fail("et_epipolar_forward_general: ATTENTION max takes no prior");
    if (int e = check_sim_prior("et_epipolar_forward_general", flags, prior != nullptr)) return e;
    const bool pool = flags & ET_GENERAL_POOLING;
    if (pool && (desc->K & 1)) return fail("et_epipolar_forward_general: POOLING needs an even K (K=%d)", desc->K);
    if ((flags & ET_GENERAL_PRIOR_MUL) && !prior) return fail("et_epipolar_forward_general: PRIOR_MUL without a prior");
    const long long hw = (long long)desc->H * desc->W;
    if (hw * (c_sim > c_val ? c_sim : c_val) * 4 >= (1LL << 31)) return fail("one feature map must stay below 2 GiB");
    GeneralParams p;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.q = q; p.m_sim = map_sim; p.m_val = map_val; p.prior = prior;
    p.out = out; p.attn = attn; p.corr = corr_pos;
    p.cs = c_sim; p.cv = c_val; p.prior_mul = (flags & ET_GENERAL_PRIOR_MUL) ? 1 : 0;
    p.attn_max = (flags & ET_GENERAL_ATTENTION_MAX) ? 1 : 0;
    p.cosine = (flags & (ET_GENERAL_COSINE | ET_GENERAL_ATTENTION_MAX)) ? 1 : 0;   // (ATTENTION max is always cosine, epipolar.py:282)
    p.sim_prior = (flags & ET_GENERAL_SIM_PRIOR) ? 1 : 0;
    const long long blocks = (hw * desc->N + kGenWaves - 1) / kGenWaves;
    if (blocks > 0x7fffffffLL) return fail("grid too large");
    const size_t lds = gen_lds_bytes(desc->K);
    hipStream_t st = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL(epipolar_fwd_general_kernel<true>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, p);
    else hipLaunchKernelGGL(epipolar_fwd_general_kernel<false>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, p);
    return check_launch("et_epipolar_forward_general");
}

int et_epipolar_backward_general(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                 const float *cam, const float *q, const float *map_sim, const float *map_val,
                                 const float *prior, const float *grad_out, int c_sim, int c_val, int flags, float *grad_q,
                                 float *grad_map_sim, float *grad_map_val, float *grad_prior, void *stream)
{
    if (!desc) return fail("et_epipolar_backward_general: desc is NULL");
    EtLayerDesc chk = *desc;
    chk.C = 4;
    if (int e = validate(&chk)) return e;
    if (!xs || !ys || !steps || !cam || !q || !map_sim || !map_val || !grad_out || !grad_q)
        return fail("et_epipolar_backward_general: NULL pointer");
    if (c_sim <= 0 || c_sim > kGenMaxQ * kWave) return fail("et_epipolar_backward_general: c_sim=%d outside [1, %d]", c_sim, kGenMaxQ * kWave);
    if (c_val <= 0 || c_val > 4096) return fail("et_epipolar_backward_general: c_val=%d outside [1, 4096]", c_val);
    if (flags & ~(ET_GENERAL_POOLING | ET_GENERAL_PRIOR_MUL | ET_GENERAL_COSINE | ET_GENERAL_ATTENTION_MAX | ET_GENERAL_SIM_PRIOR))
        return fail("et_epipolar_backward_general: unknown flag bits %d", flags);
    if ((flags & ET_GENERAL_ATTENTION_MAX) && prior && !(flags & ET_GENERAL_SIM_PRIOR))
        return fail("et_epipolar_backward_general: ATTENTION max takes no prior");
    if ((flags & ET_GENERAL_PRIOR_MUL) && !prior) return fail("et_epipolar_backward_general: PRIOR_MUL without a prior");
    if (int e = check_sim_prior("et_epipolar_backward_general", flags, prior != nullptr)) return e;
    if (grad_prior && !prior) return fail("et_epipolar_backward_general: grad_prior without a prior");
    if (desc->K > 256 * ((flags & ET_GENERAL_POOLING) ? 2 : 1)) return fail("et_epipolar_backward_general: K' = %d > 256", desc->K);
    const bool pool = flags & ET_GENERAL_POOLING;
    if (pool && (desc->K & 1)) return fail("et_epipolar_backward_general: POOLING needs an even K (K=%d)", desc->K);
    const long long hw = (long long)desc->H * desc->W;
    if (hw * (c_sim > c_val ? c_sim : c_val) * 4 >= (1LL << 31)) return fail("one feature map must stay below 2 GiB");
    GeneralBwdParams bp;
    bp.f.d = *desc;
    bp.f.xs = xs; bp.f.ys = ys; bp.f.steps = steps; bp.f.cam = cam;
    bp.f.q = q; bp.f.m_sim = map_sim; bp.f.m_val = map_val; bp.f.prior = prior;
    bp.f.out = nullptr; bp.f.attn = nullptr; bp.f.corr = nullptr;
    bp.f.cs = c_sim; bp.f.cv = c_val; bp.f.prior_mul = (flags & ET_GENERAL_PRIOR_MUL) ? 1 : 0;
    bp.f.attn_max = (flags & ET_GENERAL_ATTENTION_MAX) ? 1 : 0;
    bp.f.cosine = (flags & (ET_GENERAL_COSINE | ET_GENERAL_ATTENTION_MAX)) ? 1 : 0;
    bp.f.sim_prior = (flags & ET_GENERAL_SIM_PRIOR) ? 1 : 0;
    bp.gout = grad_out; bp.gq = grad_q; bp.gsim = grad_map_sim; bp.gval = grad_map_val; bp.gprior = grad_prior;
    const long long blocks = (hw * desc->N + kGenWaves - 1) / kGenWaves;
    if (blocks > 0x7fffffffLL) return fail("grid too large");
    const size_t lds = (size_t)kGenWaves * gen_bwd_wave_floats(desc->K) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL(epipolar_bwd_general_kernel<true>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, bp);
    else hipLaunchKernelGGL(epipolar_bwd_general_kernel<false>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, bp);
    return check_launch("et_epipolar_backward_general");
}

}  // extern "C"
