// Microbenchmark (development): cycles per buffer_load_dwordx4 wave-instruction for the access patterns of the tile
// GEMMs, data resident in L2 (4 MiB map, re-read), W waves per CU issuing loads back to back (8 in flight each).
//   pattern 0: coalesced      lane l reads 16 B at row r, byte 16 l               (1 row  x 1 KB per instruction)
//   pattern 1: fragment       lane (li, lh) reads 16 B at row R[li], byte 32 c + 16 lh    (32 rows x 32 B)
//   pattern 2: fragment64     lane (li, lh) reads 16 B at row R[li], byte 64 c + 32 lh + {0,16} (two loads: 32 rows x 64 B)
//   pattern 3: quarter rows   lane l: row R[l / 16], byte 256 c + 16 (l % 16)      (4 rows x 256 B)
//   pattern 5: 16 x 64 B      lane l: row R[l % 16], byte 64 c + 16 (l / 16)        (16 rows x 64 B: a 16x16x32 MFMA operand)
//   pattern 6: 8 x 128 B      lane l: row R[l % 8], byte 128 c + 16 (l / 8)         (8 rows x 128 B)
//   pattern 7: 16 x 64 B, quad-contiguous   lane l: row R[l / 4], byte 64 c + 16 (l % 4)
//   pattern 4: 8-byte pairs   dwordx2: lanes 0-31 row A 256 B, lanes 32-63 row B 256 B (G2 pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int PAT, int NF>
__global__ __launch_bounds__(256) void k(const float *map, const int *rows, int nrows, int iters, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(map), 0, 4096u * 1024u, 0x00020000);
    const int li = lane & 31, lh = lane >> 5;
    float acc = 0.f;
    const int base = (blockIdx.x * 4 + wave) * 37;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        u32x4 v[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int step = i * NF + u;
            int off;
            if (PAT == 0) off = rows[(base + step) % nrows] * 1024 + lane * 16;
            else if (PAT == 1) off = rows[(base + (step / 32) * 32 + li) % nrows] * 1024 + (step % 32) * 32 + lh * 16;
            else if (PAT == 2) off = rows[(base + (step / 32) * 32 + li) % nrows] * 1024 + ((step % 32) / 2) * 64 + lh * 32 + (step & 1) * 16;
            else if (PAT == 3) off = rows[(base + (step / 4) * 4 + lane / 16) % nrows] * 1024 + (step % 4) * 256 + (lane % 16) * 16;
            else if (PAT == 5) off = rows[(base + (step / 16) * 16 + lane % 16) % nrows] * 1024 + (step % 16) * 64 + (lane / 16) * 16;
            else if (PAT == 6) off = rows[(base + (step / 8) * 8 + lane % 8) % nrows] * 1024 + (step % 8) * 128 + (lane / 8) * 16;
            else if (PAT == 7) off = rows[(base + (step / 16) * 16 + lane / 4) % nrows] * 1024 + (step % 16) * 64 + (lane % 4) * 16;
            else off = rows[(base + step * 2 + lh) % nrows] * 1024 + ((step % 4) * 64 + 2 * li) * 4;
            if (PAT == 4) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(src, off, 0, 0);
                v[u] = u32x4{w.x, w.y, 0u, 0u};
            } else {
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(src, off, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) acc += __uint_as_float(v[u].x) + __uint_as_float(v[u].y);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PAT, int NF>
void run(const char *name, const float *map, const int *rows, int nrows, float *out, long long *cyc, int waves)
{
    const int iters = 16000 / NF;
    k<PAT, NF><<<256, 64 * waves>>>(map, rows, nrows, iters, out, cyc);
    hipDeviceSynchronize();
    k<PAT, NF><<<256, 64 * waves>>>(map, rows, nrows, iters, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(1024);
    hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m += h[b * 4 + w];
    m /= 256.0 * waves;
    printf("%-14s %d waves/CU, %2d loads in flight per wave: %.1f cycles of CU time per instruction, %.1f B/clk/CU\n", name, waves, NF,
           m / iters / NF / waves, (PAT == 4 ? 512.0 : 1024.0) * waves / (m / iters / NF));
}
__global__ __launch_bounds__(256) void kperm(int iters, int *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int addr = (4 * (lane & 15) + (lane >> 4)) * 4;
    int v[8];
    for (int r = 0; r < 8; ++r) v[r] = lane * 7 + r;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = __builtin_amdgcn_ds_bpermute(addr, v[r]);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
int main()
{
    {
        int *o; long long *c;
        hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 1024 * 8);
        for (int waves : {1, 4}) {
            kperm<<<256, 64 * waves>>>(2000, o, c); hipDeviceSynchronize();
            kperm<<<256, 64 * waves>>>(2000, o, c); hipDeviceSynchronize();
            std::vector<long long> h(1024);
            hipMemcpy(h.data(), c, 1024 * 8, hipMemcpyDeviceToHost);
            double m = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) m += h[b * 4 + w];
            m /= 256.0 * waves;
            printf("ds_bpermute_b32, %d waves/CU, 8 independent per wave: %.1f cycles per instruction per wave, %.1f cycles of CU time\n",
                   waves, m / 2000 / 8, m / 2000 / 8 / waves);
        }
    }
    float *map, *out; int *rows; long long *cyc;
    hipMalloc(&map, 4096u * 1024u); hipMemset(map, 0, 4096u * 1024u);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int nrows = 4096;
    std::vector<int> r(nrows);
    // rows of a tile: mostly short runs of neighbouring pixels along a line -- emulate with a stride-7 walk (distinct rows)
    for (int i = 0; i < nrows; ++i) r[i] = (i * 7 + (i / 64) * 13) % 4096;
    hipMalloc(&rows, nrows * 4); hipMemcpy(rows, r.data(), nrows * 4, hipMemcpyHostToDevice);
    for (int waves : {4}) {
        run<0, 8>("coalesced", map, rows, nrows, out, cyc, waves);
        run<0, 16>("coalesced", map, rows, nrows, out, cyc, waves);
        run<0, 32>("coalesced", map, rows, nrows, out, cyc, waves);
        run<1, 8>("fragment32", map, rows, nrows, out, cyc, waves);
        run<1, 16>("fragment32", map, rows, nrows, out, cyc, waves);
        run<1, 32>("fragment32", map, rows, nrows, out, cyc, waves);
        run<3, 8>("quarter-rows", map, rows, nrows, out, cyc, waves);
        run<3, 16>("quarter-rows", map, rows, nrows, out, cyc, waves);
        run<3, 32>("quarter-rows", map, rows, nrows, out, cyc, waves);
        run<5, 8>("16x64B", map, rows, nrows, out, cyc, waves);
        run<5, 32>("16x64B", map, rows, nrows, out, cyc, waves);
        run<6, 8>("8x128B", map, rows, nrows, out, cyc, waves);
        run<6, 32>("8x128B", map, rows, nrows, out, cyc, waves);
        run<7, 8>("16x64B quads", map, rows, nrows, out, cyc, waves);
        run<7, 32>("16x64B quads", map, rows, nrows, out, cyc, waves);
        run<4, 8>("pairs-8B", map, rows, nrows, out, cyc, waves);
        run<4, 32>("pairs-8B", map, rows, nrows, out, cyc, waves);
    }
    return 0;
}
