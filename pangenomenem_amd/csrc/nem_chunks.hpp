// nem_chunks.hpp -- device-side formation of the NEM problems of PPanGGOLiN's chunk voting loop.
//
// The reference solves a pangenome of more than 500 organisms as many NEM problems, each on a random sample of the
// organisms (`partition()`, ppanggolin.py:995-1097: `orgs = sample(organisms, chunck_size)`), and writes every sample's
// five input files from ONE graph (`__write_nem_input_files`, ppanggolin.py:821-930):
//   * the columns of the presence/absence matrix are the sampled organisms, in sample order (:850);
//   * a family with no sampled organism is dropped, the others are numbered in the master's order (:849-852);
//   * an edge's weight is the number of sampled organisms that carry the adjacency (`coverage`, :866-878); an edge
//     nobody in the sample carries is dropped, the neighbours of a family keep the master's order.
// Here the master lives on the device (organism-major bit rows, the graph in CSR, per directed edge the bit set of its
// organisms) and a problem is FORMED there: no matrix, no graph crosses PCIe per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace nemk {

// the master on the device (pointers only; nem_engine.hip owns the memory)
struct MasterDev {
    int n, d, wf, nw64, nnz;
    const uint64_t* xt;           // [d][nw64]: bit i of row o = family i present in organism o
    const int* nei_ptr;           // [n + 1]
    const int* nei_idx;           // [nnz]
    const uint32_t* edge_bits;    // [nnz][wf]: organisms that carry the directed edge
};

// per chunk, phase 1 (what decides the problem's sizes) ...
struct ChunkPlan {
    const int* organisms;         // [dc] device copy of the sample, column order
    int dc;
    uint32_t* mask;               // [wf]     the sample as a bit set of the master's organisms
    uint64_t* keep;               // [nw64]   families with at least one sampled organism
    int* list;                    // [n]      kept family j -> master index
    int* map;                     // [n]      master index -> kept number, -1: dropped
    uint16_t* cov;                // [nnz]    coverage of every directed master edge in the sample
    int* ptr;                     // [n + 1]  CSR row pointers of the chunk's graph (kept rows)
    int* counts;                  // [2]      {kept families, kept directed edges}
};
// ... and phase 2 (the engine's own buffers, filled in place)
struct ChunkFill {
    int nc, dc, wfc, npad;        // kept families, organisms of the sample, words per bit row, families padded to 256
    uint32_t* xf;                 // [nc][wfc] family-major bit rows (the engine's staging rows)
    int* perm;                    // [npad]   lane order of the density kernels: inside every 256-family tile by popcount
    int* out_ptr; int* out_idx; float* out_w;   // the engine's graph block
};

void launch_master_transpose(const uint32_t* xf, int n, int wf, int d, int nw64, uint64_t* xt, hipStream_t s);
// phase 1 for `count` chunks: plans[] is an array in DEVICE memory
void launch_chunk_plan(const MasterDev& m, const ChunkPlan* plans_dev, int count, int max_dc, hipStream_t s);
// phase 2 for one chunk
void launch_chunk_fill(const MasterDev& m, const ChunkPlan& plan, const ChunkFill& fill, hipStream_t s);
int chunk_mask_words_max();

}  // namespace nemk
