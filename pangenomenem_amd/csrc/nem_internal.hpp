// nem_internal.hpp -- internal declarations shared by the engine, the file layer and the C ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nem_mi355x.h"

namespace nemk {

constexpr double kEpsilon = 1e-20;   // EPSILON, reference nem_typ.h:63
constexpr int kMaxK = 32;            // classes supported by the kernels

void set_error(const std::string& msg);

// counter-based tie-break hash; identical to mix32() in nem_kernels.hip and orc_mix32() in the oracle
inline uint32_t mix32_host(uint32_t seed, uint32_t sweep, uint32_t site)
{
    uint32_t h = seed * 0x9E3779B1u + sweep * 0x85EBCA77u + site * 0xC2B2AE3Du + 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

// ---- file layer (nem_io.cpp): the reference's ASCII formats, SURVEY.md §5.6 -------------
struct NemInputs {
    int n = 0, d = 0;
    char type = 'S';                       // S | N   (I = image is out of scope)
    std::string str_comment;               // opening '#' comment of .str (echoed in .stderr)
    std::string nei_comment;
    std::vector<uint32_t> xbits;           // family-major bit rows, ceil(d/32) words per family
    std::vector<int32_t> nei_ptr, nei_idx; // CSR, 0-based, .nei order after ReadPtsNeighs filtering
    std::vector<float> nei_w;
    int max_neighs = 0;
    int param_mode = 1;                    // .m flag: 1 = initial values, 2 = fixed
    std::vector<float> prop, center, disp; // k, k*d, k*d
};

// each returns a StatusET-style code (NEMGPU_OK, NEMGPU_E_FILEIN, NEMGPU_E_FILE, ...)
int read_str_file(const std::string& base, NemInputs& in, std::string& err);
int read_dat_file(const std::string& base, NemInputs& in, std::string& err);
int read_nei_file(const std::string& base, NemInputs& in, std::string& err);
int read_param_file(const std::string& base, int k, NemInputs& in, std::string& err);

int write_uf_file(const std::string& path, const float* c_nk, int n, int k);
int write_cf_file(const std::string& path, const float* c_nk, int n, int k, int tie_rule, uint32_t seed, long draws_before = 0);
int write_mf_file(const std::string& path, const float crit[6], float beta, int d, int k,
                  const float* center, const float* prop, const float* disp);

}  // namespace nemk
