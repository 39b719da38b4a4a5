// nem_chunks.hip -- kernels that form the NEM problem of one organism sample ("chunk") from a master pangenome that
// lives on the device (nem_chunks.hpp; the reference's __write_nem_input_files, ppanggolin.py:821-930, for the samples
// of partition()'s voting loop, ppanggolin.py:1045-1086).  Integer / bit work only; gfx950, wave64.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>

#include "nem_chunks.hpp"

namespace nemk {

constexpr int kMaskWordsMax = 4096;          // organisms of a master / 32 (131 072 organisms)

// exclusive scan of one int per thread over a 1024-thread block; returns this thread's offset, *total = the block's sum
__device__ __forceinline__ int block_scan_1024(int v, int* total)
{
    __shared__ int s_wave[16];
    __shared__ int s_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < 16; w++) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_total = run;
    }
    __syncthreads();
    const int off = s_wave[wave] + incl - v;
    *total = s_total;
    __syncthreads();                                     // (the arrays are free for the next call)
    return off;
}

// family-major bit rows [n][wf] -> organism-major bit rows [d][nw64] (bit i of row o = family i present in organism o)
__global__ __launch_bounds__(64) void k_master_transpose(const uint32_t* __restrict__ xf, int n, int wf, int d, int nw64,
                                                        uint64_t* __restrict__ xt)
{
    const int g = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    const int i = g * 64 + lane;
    const uint32_t x = i < n ? xf[(size_t)i * wf + w] : 0u;
    uint64_t mine = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) {
        const uint64_t bal = __ballot((x >> b) & 1u);
        if (lane == b) mine = bal;
    }
    const int o = w * 32 + lane;
    if (lane < 32 && o < d) xt[(size_t)o * nw64 + g] = mine;
}

// the sample as a bit set of the master's organisms; the chunk's counters cleared
__global__ __launch_bounds__(256) void k_chunk_mask(const ChunkPlan* __restrict__ plans, int wf)
{
    __shared__ uint32_t s_mask[kMaskWordsMax];
    const ChunkPlan p = plans[blockIdx.x];
    for (int w = threadIdx.x; w < wf; w += 256) s_mask[w] = 0u;
    __syncthreads();
    for (int t = threadIdx.x; t < p.dc; t += 256) { const int o = p.organisms[t]; atomicOr(&s_mask[o >> 5], 1u << (o & 31)); }
    __syncthreads();
    for (int w = threadIdx.x; w < wf; w += 256) p.mask[w] = s_mask[w];
    if (threadIdx.x < 2) p.counts[threadIdx.x] = 0;
}

// families present in at least one sampled organism (ppanggolin.py:849: `if not organisms.isdisjoint(node_organisms)`)
__global__ __launch_bounds__(256) void k_chunk_keep(const ChunkPlan* __restrict__ plans, const uint64_t* __restrict__ xt, int nw64)
{
    const ChunkPlan p = plans[blockIdx.y];
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= nw64) return;
    uint64_t acc = 0;
    for (int t = 0; t < p.dc; t++) acc |= xt[(size_t)p.organisms[t] * nw64 + g];
    p.keep[g] = acc;
}

// the kept families numbered in the master's order (index_fam, ppanggolin.py:851): list[j] = master index, map[i] = j | -1
__global__ __launch_bounds__(1024) void k_chunk_index(const ChunkPlan* __restrict__ plans, int n, int nw64)
{
    const ChunkPlan p = plans[blockIdx.x];
    int base = 0;
    for (int w0 = 0; w0 < nw64; w0 += 1024) {
        const int w = w0 + threadIdx.x;
        const uint64_t v = w < nw64 ? p.keep[w] : 0ull;
        int total = 0;
        int pos = base + block_scan_1024((int)__popcll(v), &total);
        if (w < nw64) {
            for (int b = 0; b < 64; b++) {
                const int i = w * 64 + b;
                if (i >= n) break;
                if ((v >> b) & 1ull) { p.list[pos] = i; p.map[i] = pos; pos++; }
                else p.map[i] = -1;
            }
        }
        base += total;
    }
    if (threadIdx.x == 0) p.counts[0] = base;
}

// coverage of every directed master edge in the sample: the organisms of the sample that carry it (ppanggolin.py:866-876)
__global__ __launch_bounds__(256) void k_chunk_cov(const ChunkPlan* __restrict__ plans, const uint32_t* __restrict__ edge_bits, int nnz, int wf)
{
    __shared__ uint32_t s_mask[kMaskWordsMax];
    const ChunkPlan p = plans[blockIdx.y];
    for (int w = threadIdx.x; w < wf; w += 256) s_mask[w] = p.mask[w];
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t* __restrict__ row = edge_bits + (size_t)e * wf;
    int c = 0;
    for (int w = 0; w < wf; w++) c += __popc(row[w] & s_mask[w]);
    p.cov[e] = (uint16_t)(c > 65535 ? 65535 : c);
}

// an edge of the chunk's graph: carried by a sampled organism (`if coverage == 0: continue`, ppanggolin.py:877), between
// two kept families (always so when the edge sets agree with the matrix; an edge to a dropped family is dropped too)
__device__ __forceinline__ bool chunk_edge(const ChunkPlan& p, const int* __restrict__ nei_idx, int e)
{
    return p.cov[e] != 0 && p.map[nei_idx[e]] >= 0;
}

// row pointers of the chunk's graph over the kept families
__global__ __launch_bounds__(1024) void k_chunk_ptr(const ChunkPlan* __restrict__ plans, const int* __restrict__ nei_ptr,
                                                   const int* __restrict__ nei_idx)
{
    const ChunkPlan p = plans[blockIdx.x];
    const int nc = p.counts[0];
    int base = 0;
    for (int j0 = 0; j0 < nc; j0 += 1024) {
        const int j = j0 + threadIdx.x;
        int deg = 0;
        if (j < nc) {
            const int i = p.list[j];
            for (int e = nei_ptr[i]; e < nei_ptr[i + 1]; e++) deg += chunk_edge(p, nei_idx, e) ? 1 : 0;
        }
        int total = 0;
        const int off = block_scan_1024(deg, &total);
        if (j < nc) p.ptr[j] = base + off;
        base += total;
    }
    if (threadIdx.x == 0) { p.ptr[nc] = base; p.counts[1] = base; }
}

// ---- phase 2: the engine's own buffers
// One block per 256-family tile: the tile's bit rows (column t = organism organisms[t]: ppanggolin.py:850) and, from
// their popcounts, the tile's lane order for the density kernels (stable by popcount, as upload_bits sorts on the host).
// The tile's 256 kept families are consecutive set bits of the master's order: they span a few 64-family words of every
// organism row.  Those words go through LDS, kChunkOrgs organisms at a time (coalesced loads, one per thread and word),
// and every thread picks its family's bit out of them -- instead of one global load per family and organism.
constexpr int kChunkOrgs = 256;              // organisms staged per pass
constexpr int kChunkSpan = 12;               // 64-family words of a row a pass can stage (768 master families under the tile)
__global__ __launch_bounds__(256) void k_chunk_rows(ChunkPlan p, ChunkFill f, const uint64_t* __restrict__ xt, int nw64)
{
    __shared__ int s_pc[256];
    __shared__ uint64_t s_rows[kChunkOrgs][kChunkSpan + 1];
    const int tile0 = blockIdx.x * 256;
    const int j = tile0 + threadIdx.x;
    const int jlast = min(tile0 + 255, f.nc - 1);
    const int g0 = p.list[tile0] >> 6, g1 = p.list[jlast] >> 6;   // (tile0 < nc: the grid has a block per started tile)
    const bool staged = g1 - g0 < kChunkSpan;
    const int i = j < f.nc ? p.list[j] : 0;
    const int g = i >> 6, bit = i & 63;
    int pc = j < f.nc ? 0 : 0x7fffffff;
    for (int t0 = 0; t0 < f.dc; t0 += kChunkOrgs) {              // (kChunkOrgs is a multiple of 32: passes start on a word)
        const int tn = min(kChunkOrgs, f.dc - t0);
        if (staged) {
            __syncthreads();
            if ((int)threadIdx.x < tn) {
                const uint64_t* __restrict__ row = xt + (size_t)p.organisms[t0 + threadIdx.x] * nw64 + g0;
                for (int q = 0; q <= g1 - g0; q++) s_rows[threadIdx.x][q] = row[q];
            }
            __syncthreads();
        }
        if (j < f.nc) {
            for (int w = 0; 32 * w < tn; w++) {
                uint32_t word = 0;
                const int b1 = min(32, tn - 32 * w);
                for (int b = 0; b < b1; b++) {
                    const uint64_t v = staged ? s_rows[32 * w + b][g - g0] : xt[(size_t)p.organisms[t0 + 32 * w + b] * nw64 + g];
                    word |= (uint32_t)((v >> bit) & 1ull) << b;
                }
                f.xf[(size_t)j * f.wfc + (t0 >> 5) + w] = word;
                pc += __popc(word);
            }
        }
    }
    s_pc[threadIdx.x] = pc;
    __syncthreads();
    if (j < f.nc) {
        int rank = 0;
        for (int q = 0; q < 256; q++) {
            const int o = s_pc[q];
            rank += (o < pc || (o == pc && q < (int)threadIdx.x)) ? 1 : 0;
        }
        f.perm[tile0 + rank] = j;
    } else if (j < f.npad) f.perm[j] = j;
}

// the chunk's graph in the engine's CSR block: neighbours in the master's order, weights = coverage
__global__ __launch_bounds__(256) void k_chunk_graph(ChunkPlan p, ChunkFill f, const int* __restrict__ nei_ptr, const int* __restrict__ nei_idx)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) f.out_ptr[f.nc] = p.ptr[f.nc];
    if (j >= f.nc) return;
    const int i = p.list[j];
    int pos = p.ptr[j];
    f.out_ptr[j] = pos;
    for (int e = nei_ptr[i]; e < nei_ptr[i + 1]; e++) {
        if (chunk_edge(p, nei_idx, e)) { f.out_idx[pos] = p.map[nei_idx[e]]; f.out_w[pos] = (float)p.cov[e]; pos++; }
    }
}

void launch_master_transpose(const uint32_t* xf, int n, int wf, int d, int nw64, uint64_t* xt, hipStream_t s)
{
    hipLaunchKernelGGL(k_master_transpose, dim3(nw64, wf), dim3(64), 0, s, xf, n, wf, d, nw64, xt);
}

void launch_chunk_plan(const MasterDev& m, const ChunkPlan* plans_dev, int count, int max_dc, hipStream_t s)
{
    (void)max_dc;
    hipLaunchKernelGGL(k_chunk_mask, dim3(count), dim3(256), 0, s, plans_dev, m.wf);
    hipLaunchKernelGGL(k_chunk_keep, dim3((m.nw64 + 255) / 256, count), dim3(256), 0, s, plans_dev, m.xt, m.nw64);
    hipLaunchKernelGGL(k_chunk_index, dim3(count), dim3(1024), 0, s, plans_dev, m.n, m.nw64);
    if (m.nnz > 0) hipLaunchKernelGGL(k_chunk_cov, dim3((m.nnz + 255) / 256, count), dim3(256), 0, s, plans_dev, m.edge_bits, m.nnz, m.wf);
    hipLaunchKernelGGL(k_chunk_ptr, dim3(count), dim3(1024), 0, s, plans_dev, m.nei_ptr, m.nei_idx);
}

void launch_chunk_fill(const MasterDev& m, const ChunkPlan& plan, const ChunkFill& fill, hipStream_t s)
{
    hipLaunchKernelGGL(k_chunk_rows, dim3(fill.npad / 256), dim3(256), 0, s, plan, fill, m.xt, m.nw64);
    hipLaunchKernelGGL(k_chunk_graph, dim3((fill.nc + 255) / 256), dim3(256), 0, s, plan, fill, m.nei_ptr, m.nei_idx);
}

int chunk_mask_words_max() { return kMaskWordsMax; }

}  // namespace nemk
