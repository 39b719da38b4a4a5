// nem_sweep.hip -- every instance of the E2 relaxation round (k_sweep, its batched twin) and their launch wrapper.
// The round itself is sweep_body in nem_sweep_dev.hpp (ComputePartitionNEM / ComputeLocalProba / ComputeMAP,
// nem_alg.c:2330-2405, 2546-2616, 590-645).  A translation unit of its own: the instances are a third of the library's
// compile time, and build.py compiles the units side by side.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "nem_kernels.hpp"
#include "nem_sweep_dev.hpp"

namespace nemk {

template <int KT, bool NCEM, int BS, bool LIBC = false>
__global__ __launch_bounds__(BS) void k_sweep(SweepArgs a) { sweep_body<KT, NCEM, BS, LIBC>(a, blockIdx.x, gridDim.x); }
template <int KT, bool NCEM, int BS, bool LIBC = false>
__global__ __launch_bounds__(BS) void k_sweep_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(SweepArgs) sweep_body<KT, NCEM, BS, LIBC>(a, blockIdx.x, nblk); }

// up to SweepArgs::fused_rounds relaxation rounds in one launch (NCEM, hash / first tie rule, one engine, at most
// kFusedMaxBlocks blocks: every block must be resident, they meet between the rounds)
template <int KT, int BS>
__global__ __launch_bounds__(BS) void k_sweep_fused(SweepArgs a) { sweep_body<KT, true, BS, false, true>(a, blockIdx.x, gridDim.x); }

// SweepArgs::exp_tab: exp((double)beta * (double)(float)m) by the exp the sweep kernels call on the same argument
__global__ void k_exp_table(float beta, double* tab, int len)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < len) tab[m] = exp((double)beta * (double)(float)m);
}
void launch_exp_table(float beta, double* tab, int len, hipStream_t s)
{
    hipLaunchKernelGGL(k_exp_table, dim3((len + 255) / 256), dim3(256), 0, s, beta, tab, len);
}

static int sweep_variant(int K, bool ncem, bool big, bool libc) { return (K >= 1 && K <= 10 ? K : 0) | (ncem ? 16 : 0) | (big ? 32 : 0) | (libc && ncem ? 64 : 0); }

template <bool BATCHED>
static void sweep_dispatch(int variant, dim3 grid, unsigned bdim, hipStream_t s, const SweepArgs* a, const void* arr, int stride, const int* gx)
{
    const int kt = variant & 15; const bool ncem = (variant & 16) != 0, big = (variant & 32) != 0, libc = (variant & 64) != 0;
    dim3 block(bdim);                                    // (256; the large-shard instances: SweepArgs::spb rounded up to whole waves, at most 1024)
#define NEM_SW2(KT_, NC_, BS_, LC_)                                                                                \
    do {                                                                                                           \
        if (BATCHED) hipLaunchKernelGGL((k_sweep_b<KT_, NC_, BS_, LC_>), grid, block, 0, s, arr, stride, gx);     \
        else hipLaunchKernelGGL((k_sweep<KT_, NC_, BS_, LC_>), grid, block, 0, s, *a);                             \
    } while (0)
#define NEM_SW(KT_)                                                                                                \
    case KT_:                                                                                                      \
        if (big) { if (!ncem) NEM_SW2(KT_, false, 1024, false); else if (libc) NEM_SW2(KT_, true, 1024, true); else NEM_SW2(KT_, true, 1024, false); } \
        else { if (!ncem) NEM_SW2(KT_, false, 256, false); else if (libc) NEM_SW2(KT_, true, 256, true); else NEM_SW2(KT_, true, 256, false); }       \
        break;
    switch (kt) {
        NEM_SW(1) NEM_SW(2) NEM_SW(3) NEM_SW(4) NEM_SW(5) NEM_SW(6) NEM_SW(7) NEM_SW(8) NEM_SW(9) NEM_SW(10)
    default:
        if (!ncem) NEM_SW2(0, false, 256, false); else if (libc) NEM_SW2(0, true, 256, true); else NEM_SW2(0, true, 256, false);
    }
#undef NEM_SW
#undef NEM_SW2
}

// the launch geometry of a round: block size (= sites per block) for n_local sites
static int sweep_block_sites(int n_local, int K, bool* big_out)
{
    const bool generic = !(K >= 1 && K <= 10);
    const bool big = n_local >= 65536 && !generic;
    static const bool balanced = !(getenv("NEM_MI355X_SWEEP_SPB") && getenv("NEM_MI355X_SWEEP_SPB")[0] == '0');   // (0: 1024 sites per block)
    int bs = 256;
    if (big && !balanced) bs = 1024;
    else if (big) {
        const int waves_of_blocks = (n_local + 256 * 1024 - 1) / (256 * 1024);
        const int per_block = (n_local + 256 * waves_of_blocks - 1) / (256 * waves_of_blocks);
        bs = std::min(1024, std::max(256, (per_block + 63) / 64 * 64));
    }
    if (big_out) *big_out = big;
    return bs;
}
int sweep_grid_blocks(int n_local, int K)
{
    const int bs = sweep_block_sites(n_local, K, nullptr);
    return (n_local + bs - 1) / bs;
}

static unsigned long long* g_prof_dev = nullptr;
int sweep_phases_read(unsigned long long* out64)
{
    if (g_prof_dev == nullptr) return -1;
    return hipMemcpy(out64, g_prof_dev, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

static bool launch_sweep_fused(const SweepArgs& a0, int bs, bool big, dim3 grid, hipStream_t s)
{
    SweepArgs a = a0;
    static const bool prof = getenv("NEM_MI355X_SWEEP_PROF") && getenv("NEM_MI355X_SWEEP_PROF")[0] == '1';
    if (prof && g_prof_dev == nullptr) { if (hipMalloc(&g_prof_dev, 64 * sizeof(unsigned long long)) != hipSuccess) g_prof_dev = nullptr; }
    a.prof = prof ? g_prof_dev : nullptr;
    if (a.prof != nullptr) (void)hipMemsetAsync(a.prof, 0, 64 * sizeof(unsigned long long), s);
#define NEM_SF(KT_) case KT_:                                                                     \
        if (big) hipLaunchKernelGGL((k_sweep_fused<KT_, 1024>), grid, dim3(bs), 0, s, a);        \
        else hipLaunchKernelGGL((k_sweep_fused<KT_, 256>), grid, dim3(bs), 0, s, a);             \
        return true;
    switch (a.K) { NEM_SF(2) NEM_SF(3) NEM_SF(4) NEM_SF(5) NEM_SF(6) NEM_SF(7) NEM_SF(8) NEM_SF(9) NEM_SF(10) default: return false; }
#undef NEM_SF
}
bool sweep_fused_has_instance(int K) { return K >= 2 && K <= 10; }

void launch_sweep(const SweepArgs& a0, bool ncem, hipStream_t s)
{
    // Large shards run up to 1024 sites per block (the per-block flag atomics -- one address -- were what a round cost
    // when every site reports something, e.g. all densities underflow at D = 5000; the tally now rides on the
    // last-block ticket, but a round is still bound by each block's instruction stream): as many sites per block as
    // fill the chip's 256 CUs evenly -- 200 000 sites: 241 blocks of 832 instead of 196 of 1024 on 256 CUs.
    bool big = false;
    SweepArgs a = a0;
    const int bs = sweep_block_sites(a.n_local, a.K, &big);
    a.spb = big ? bs : 0;
    dim3 grid((a.n_local + bs - 1) / bs);
    const int variant = sweep_variant(a.K, ncem, big, a.tie_rule == NEMGPU_TIE_LIBC);
    if (a.fused_rounds >= 2) {                           // (the engine asks for it only where sweep_fused_ok() holds)
        if (current_recorder() == nullptr && ncem && (int)grid.x <= kFusedMaxBlocks && launch_sweep_fused(a, bs, big, grid, s)) return;
        a.fused_rounds = 0;                              // not reached: the engine tests the same conditions
    }
    if (record_op(OP_SWEEP, variant, grid, (unsigned)bs, a)) return;
    sweep_dispatch<false>(variant, grid, (unsigned)bs, s, &a, nullptr, 0, nullptr);
}


void sweep_dispatch_batched(int variant, dim3 grid, unsigned bdim, hipStream_t s, const void* arr, int stride, const int* gx)
{
    sweep_dispatch<true>(variant, grid, bdim, s, nullptr, arr, stride, gx);
}

}  // namespace nemk
