// libepipolar_amd.so: the backward for any shape (et_epipolar_backward).
#include "et_common.h"

namespace {
#include "kernels_sample_table.inc"
#include "kernels_backward.inc"  // epipolar_bwd_kernel, epipolar_bwd_emit_kernel, bwd_scan/bucket, epipolar_bwd_gather_kernel

template <int CPD, int KPL>
void launch_bwd(const BwdParams &p, int variant, dim3 grid, hipStream_t st)
{
    if (variant & ET_VARIANT_SAFE_REDUCE)
        hipLaunchKernelGGL((epipolar_bwd_kernel<CPD, KPL, false>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((epipolar_bwd_kernel<CPD, KPL, true>), grid, dim3(256), 0, st, p);
}
}  // namespace

extern "C" {

size_t et_epipolar_backward_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc)) return 0;
    const size_t rows = (size_t)desc->N * desc->H * desc->W;
    const size_t cap = 4u * (size_t)desc->K;
    return rows * cap * (3u * 4u + 16u) + rows * 4u * 4u + 256u;
}

int et_epipolar_backward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                         const float *cam, const float *feat_ref, const float *feat_src,
                         const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !grad_out || !grad_ref || !grad_src)
        return fail("et_epipolar_backward: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int HW = desc->H * desc->W;
    const size_t rows = (size_t)desc->N * HW;
    const size_t bytes = rows * desc->C * sizeof(float);
    const bool gather = workspace != nullptr && !(desc->variant & ET_VARIANT_BWD_ATOMIC) && desc->src_grad_mask != 0;
    if (gather && workspace_bytes < et_epipolar_backward_workspace_bytes(desc))
        return fail("et_epipolar_backward: workspace of %zu bytes is smaller than the %zu required", workspace_bytes,
                    et_epipolar_backward_workspace_bytes(desc));
    if (gather && ((long long)HW * 4 * desc->K >= (1LL << 31)))
        return fail("et_epipolar_backward: H*W*4K must stay below 2^31 for the gather-form backward");
    BwdParams p;
    std::memset(&p, 0, sizeof(p));
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src; p.gout = grad_out;
    p.gref = grad_ref; p.gsrc = grad_src;
    p.blocks_per_pair = (HW + kPixPerBlock - 1) / kPixPerBlock;
    const long long total = (long long)p.blocks_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    int *row_base = nullptr, *row_cursor = nullptr;
    int4 *csr = nullptr;
    if (gather) {
        // carve the workspace: 3 pixel-major entry arrays, 3 row-major (CSR) arrays, 4 per-row int arrays
        p.cap = 4 * desc->K;
        const size_t slots = rows * p.cap;
        char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
        p.ent_u = reinterpret_cast<int *>(w);        w += slots * 4;
        p.ent_a = reinterpret_cast<float *>(w);      w += slots * 4;
        p.ent_b = reinterpret_cast<float *>(w);      w += slots * 4;
        csr = reinterpret_cast<int4 *>(w);           w += slots * 16;
        p.ent_count = reinterpret_cast<int *>(w);    w += rows * 4;
        p.row_count = reinterpret_cast<int *>(w);    w += rows * 4;
        row_base = reinterpret_cast<int *>(w);       w += rows * 4;
        row_cursor = reinterpret_cast<int *>(w);
        hipError_t me = hipMemsetAsync(p.row_count, 0, rows * 4, st);
        if (me != hipSuccess) return fail("hipMemsetAsync(row_count): %s", hipGetErrorString(me));
    } else {
        hipError_t me = hipMemsetAsync(grad_src, 0, bytes, st);
        if (me != hipSuccess) return fail("hipMemsetAsync(grad_src): %s", hipGetErrorString(me));
    }
    const dim3 grid((unsigned)total);
    const int cpd = (desc->C + 63) / 64, kpl = (desc->K + 63) / 64;
    const int v = desc->variant;
    if (gather) {
        const size_t lds_e = (size_t)kWavesPerBlock * (kpl * kWave * 16 + 3 * p.cap * 4);
        const bool safe = v & ET_VARIANT_SAFE_REDUCE;
#define ET_EMIT(CPLv, KPLv)                                                                                        \
    do {                                                                                                           \
        if (safe) hipLaunchKernelGGL((epipolar_bwd_emit_kernel<CPLv, KPLv, false>), grid, dim3(256), lds_e, st, p); \
        else hipLaunchKernelGGL((epipolar_bwd_emit_kernel<CPLv, KPLv, true>), grid, dim3(256), lds_e, st, p);       \
    } while (0)
        if (desc->C <= 256) { if (kpl == 1) ET_EMIT(1, 1); else if (kpl == 2) ET_EMIT(1, 2); else ET_EMIT(1, 4); }
        else { if (kpl == 1) ET_EMIT(2, 1); else if (kpl == 2) ET_EMIT(2, 2); else ET_EMIT(2, 4); }
#undef ET_EMIT
    } else {
#define ET_BWD_CASE(CPD)                                   \
    if (kpl == 1) launch_bwd<CPD, 1>(p, v, grid, st);      \
    else if (kpl == 2) launch_bwd<CPD, 2>(p, v, grid, st); \
    else launch_bwd<CPD, 4>(p, v, grid, st);
    if (cpd <= 1) { ET_BWD_CASE(1) }
    else if (cpd <= 2) { ET_BWD_CASE(2) }
    else if (cpd <= 4) { ET_BWD_CASE(4) }
    else { ET_BWD_CASE(8) }
#undef ET_BWD_CASE
    }
    if (int e = check_launch("et_epipolar_backward")) return e;
    if (gather) {
        hipLaunchKernelGGL(bwd_scan_kernel, dim3(desc->N), dim3(256), 0, st, HW, p.row_count, row_base, row_cursor);
        const unsigned bblocks = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock < 16384
                                            ? (rows + kWavesPerBlock - 1) / kWavesPerBlock : 16384);
        hipLaunchKernelGGL(bwd_bucket_kernel, dim3(bblocks), dim3(256), 0, st, HW, p.cap, (int)rows, p.ent_count,
                           p.ent_u, p.ent_a, p.ent_b, row_base, row_cursor, csr);
        // entries per source pixel ordered in LDS (rows with more entries: key range by key range, the same fixed order;
        // ET_VARIANT_BWD_UNSORTED: arrival order)
        const int max_sort = (desc->variant & ET_VARIANT_BWD_UNSORTED) ? 0 : 1024;
        const unsigned gblocks = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
        const size_t lds = (size_t)kWavesPerBlock * 2 * (max_sort ? max_sort : 1) * sizeof(int);
        if (desc->C <= 256)
            hipLaunchKernelGGL((epipolar_bwd_gather_kernel<1>), dim3(gblocks), dim3(256), lds, st, HW, desc->C, p.cap,
                               (int)rows, desc->src_grad_mask, p.row_count, row_base, csr, feat_ref, grad_out, grad_src,
                               max_sort);
        else
            hipLaunchKernelGGL((epipolar_bwd_gather_kernel<2>), dim3(gblocks), dim3(256), lds, st, HW, desc->C, p.cap,
                               (int)rows, desc->src_grad_mask, p.row_count, row_base, csr, feat_ref, grad_out, grad_src,
                               max_sort);
        if (int e = check_launch("et_epipolar_backward(gather)")) return e;
    }
    return 0;
}

}  // extern "C"
