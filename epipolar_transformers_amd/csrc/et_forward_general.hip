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
// not tuned further -- these modes are not on BASELINE.json's metric.  Forward only: training of these modes runs the
// chunked torch restatement (autograd).
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
};

constexpr int kGenWaves = 4;        // waves (= reference pixels) per block
constexpr int kGenMaxQ = 8;         // query channels a lane keeps in registers: cs <= 512
constexpr int gen_wave_floats(int K) { return (K * 9 + 3) & ~3; }   // 36 bytes per sample, 16-byte aligned per wave
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
    {
        const float *qrow = p.q + ((size_t)n * HW + pix) * p.cs;
        const float *m1 = p.m_sim + (size_t)n * HW * p.cs;
        float qv[kGenMaxQ];
#pragma unroll
        for (int i = 0; i < kGenMaxQ; ++i) qv[i] = (lane + i * kWave < p.cs) ? qrow[lane + i * kWave] : 0.f;
        for (int k = 0; k < Ks; ++k) {
            const int4 t0 = s_tap[k];
            const float4 w0 = s_w[k];
            int4 t1 = t0;
            float4 w1 = w0;
            if (POOL) {
                t1 = s_tap[k + Ks];
                w1 = s_w[k + Ks];
            }
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < kGenMaxQ; ++i) {
                const int c = lane + i * kWave;
                if (c < p.cs) {
                    float v = gen_sample(m1, p.cs, c, t0, w0);
                    if (POOL) v = fmaxf(v, gen_sample(m1, p.cs, c, t1, w1));
                    dot = fmaf(qv[i], v, dot);
                }
            }
            dot = wave_all_sum(dot);
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
            float v = in ? s_sim[k] : 0.f;
            v = (v == 0.f) ? -1e10f : v;                                          // epipolar.py:298
            pr[s] = (in && p.prior) ? p.prior[((size_t)n * Ks + k) * HW + pix] : 0.f;
            if (p.prior && !p.prior_mul) v += pr[s];                              // :300-301
            v = d.softmax_enabled ? v * d.softmax_scale : v / (float)Ks;          // :306 / :311
            l[s] = v;
            if (in) vmax = fmaxf(vmax, v);
        }
        if (d.softmax_enabled) {
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
                s_sim[k] = a[s];
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
    if (flags & ~(ET_GENERAL_POOLING | ET_GENERAL_PRIOR_MUL)) return fail("et_epipolar_forward_general: unknown flag bits %d", flags);
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
    const long long blocks = (hw * desc->N + kGenWaves - 1) / kGenWaves;
    if (blocks > 0x7fffffffLL) return fail("grid too large");
    const size_t lds = gen_lds_bytes(desc->K);
    hipStream_t st = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL(epipolar_fwd_general_kernel<true>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, p);
    else hipLaunchKernelGGL(epipolar_fwd_general_kernel<false>, dim3((unsigned)blocks), dim3(kWave * kGenWaves), lds, st, p);
    return check_launch("et_epipolar_forward_general");
}

}  // extern "C"
