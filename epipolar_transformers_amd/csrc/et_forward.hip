// libepipolar_amd.so: the fused forward for any shape (et_epipolar_forward).
#include "et_common.h"

namespace {
#include "kernels_sample_table.inc"
#include "kernels_forward.inc"   // SampleTable, epipolar_fwd_kernel, epipolar_fwd_multi_kernel

template <int CPL, int KPL>
void launch_fwd(const FwdParams &p, int variant, dim3 grid, size_t lds, hipStream_t st)
{
    const bool safe = variant & ET_VARIANT_SAFE_REDUCE, nocache = variant & ET_VARIANT_NO_TAP_CACHE;
    const bool b4 = variant & ET_VARIANT_BATCH4;
    const int occ = (variant & ET_VARIANT_OCC6) ? 6 : (variant & ET_VARIANT_OCC5) ? 5 : 1;
    const bool ragged = (p.d.K % 8) != 0;   // K % 4 == 0 but % 8 != 0 also takes the ragged build (fewer variants)
#define ET_FWD(B, F, Cc, W)                                                                                  \
    do {                                                                                                     \
        if (ragged)                                                                                          \
            hipLaunchKernelGGL((epipolar_fwd_kernel<CPL, KPL, B, F, Cc, W, true>), grid, dim3(256), lds, st, p);  \
        else                                                                                                 \
            hipLaunchKernelGGL((epipolar_fwd_kernel<CPL, KPL, B, F, Cc, W, false>), grid, dim3(256), lds, st, p); \
    } while (0)
    if (safe || nocache) {
        // ablation / fallback variants, default register budget
        if (safe && !nocache) { if (b4) ET_FWD(4, false, true, 1); else ET_FWD(8, false, true, 1); }
        else if (!safe && nocache) ET_FWD(8, true, false, 1);
        else ET_FWD(8, false, false, 1);
    } else if (b4) {
        if (occ == 6) ET_FWD(4, true, true, 6);
        else if (occ == 5) ET_FWD(4, true, true, 5);
        else ET_FWD(4, true, true, 1);
    } else {
        if (occ == 6) ET_FWD(8, true, true, 6);
        else if (occ == 5) ET_FWD(8, true, true, 5);
        else ET_FWD(8, true, true, 1);
    }
#undef ET_FWD
}
}  // namespace

extern "C" {

int et_epipolar_forward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                        const float *cam, const float *feat_ref, const float *feat_src, float *out,
                        float *attn, float *corr_pos, const float *res_bias, float *res_base, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !out)
        return fail("et_epipolar_forward: NULL pointer");
    if (res_bias && !res_base) return fail("et_epipolar_forward: res_bias given without res_base");
    FwdParams p;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src;
    p.out = out; p.attn = attn; p.corr = corr_pos;
    p.res_bias = res_bias; p.res_base = res_base;
    const int HW = desc->H * desc->W;
    p.blocks_per_pair = (HW + kPixPerBlock - 1) / kPixPerBlock;
    const long long total = (long long)p.blocks_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    p.interleave = (desc->variant & ET_VARIANT_PIXEL_INTERLEAVE) ? 1 : 0;
    p.ablate = 0;
    const dim3 grid((unsigned)total);
    const int kpl_ = (desc->K + 63) / 64;
    const size_t lds = (attn ? (size_t)desc->K * kPixPerBlock * sizeof(float) : 0) +
                       (size_t)kWavesPerBlock * kpl_ * kWave * 4 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int cpl = (desc->C + 255) / 256, kpl = (desc->K + 63) / 64;
    // variant 0 = the tuned default (measured on MI355X, profiles/): for the 256-channel head with K <= 64
    // four pixels per wave in lockstep; otherwise one pixel per wave, batches of 4 samples, <= 96 VGPRs (5 waves/SIMD),
    // waves of a block interleaved over neighbouring pixels
    int v = desc->variant & ~(ET_VARIANT_NO_TILE | ET_VARIANT_TILE_SPLIT);
#ifdef ET_DEV_ABLATE
    constexpr int kAblateBits = ET_VARIANT_ABLATE_NO_LOADS | ET_VARIANT_ABLATE_ONE_ROW;
#else
    constexpr int kAblateBits = 0;   // (the roofline ablations -- wrong results by construction -- are not in product builds)
#endif
    if ((v & ~kAblateBits) == 0)
        v |= (desc->C == 256 && kpl == 1) ? ET_VARIANT_MULTI4   // K > 64: its LDS records cut occupancy (measured 1.8x slower)
                                          : (ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_PIXEL_INTERLEAVE);
    if (v & ET_VARIANT_BASELINE)
        v &= ~(ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_OCC6 | ET_VARIANT_PIXEL_INTERLEAVE |
               ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4);
#ifdef ET_DEV_ABLATE
    p.ablate = (v & ET_VARIANT_ABLATE_NO_LOADS) ? 1 : (v & ET_VARIANT_ABLATE_ONE_ROW) ? 2 : 0;
#endif
    p.interleave = (v & ET_VARIANT_PIXEL_INTERLEAVE) ? 1 : 0;
    if ((v & (ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4)) && desc->C == 256 && kpl <= 2) {
        // several pixels per wave; per-wave LDS: PPW * KP * 32 + PPW * 16 bytes
        const int ppw = (v & ET_VARIANT_MULTI4) ? 4 : 2;
        const size_t lds_m = (attn ? (size_t)desc->K * kPixPerBlock * sizeof(float) : 0) +
                             (size_t)kWavesPerBlock * (ppw * kpl * kWave * 32 + ppw * 16);
        const bool occ4 = v & ET_VARIANT_OCC5;   // multi kernels: compile for 4 waves per SIMD (128 VGPRs)
        const bool pipe = v & ET_VARIANT_PIPELINE;
#define ET_MULTI(P, Q, KK)                                                                                     \
    do {                                                                                                       \
        if (pipe) hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, true, 1>), grid, dim3(256), lds_m, st, p);   \
        else if (occ4) hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, false, 4>), grid, dim3(256), lds_m, st, p); \
        else hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, false, 1>), grid, dim3(256), lds_m, st, p);    \
    } while (0)
        if (ppw == 4) { if (kpl == 1) ET_MULTI(4, 4, 1); else ET_MULTI(4, 4, 2); }
        else { if (kpl == 1) ET_MULTI(2, 2, 1); else ET_MULTI(2, 2, 2); }
#undef ET_MULTI
        return check_launch("et_epipolar_forward(multi)");
    }
    if (cpl == 1) {
        if (kpl == 1) launch_fwd<1, 1>(p, v, grid, lds, st);
        else if (kpl == 2) launch_fwd<1, 2>(p, v, grid, lds, st);
        else launch_fwd<1, 4>(p, v, grid, lds, st);
    } else {
        if (kpl == 1) launch_fwd<2, 1>(p, v, grid, lds, st);
        else if (kpl == 2) launch_fwd<2, 2>(p, v, grid, lds, st);
        else launch_fwd<2, 4>(p, v, grid, lds, st);
    }
    return check_launch("et_epipolar_forward");
}

}  // extern "C"
