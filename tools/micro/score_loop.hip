// The row loop of the scoring kernels (score_body, filmyou-core_amd/csrc/fy_rm2_kernels.hpp) in two forms, outside the library:
//
//   v1  the loop as round 4 ships it: per rated item j of the user three v_readlane (idx, e, q), the row address by a 64-bit VALU
//       multiply-add, 12 bytes per lane = four 24-bit matrix entries (v_perm + shift each), x = G + a e + q b by two scalar FMAs per
//       term, the product form of the log sum with v_frexp_mant / v_frexp_exp / integer add PER TERM, the "already rated" mask test
//       per row.  Counted in profiles/r4/k50_n1000_pmc_sq_issue.txt: 10.9 VALU instructions per log term at 66 - 91 % VALU-busy.
//   v2  the same arithmetic with the instructions that do not have to be per term taken out:
//         * idx / e / q of a batch of eight rows by wave-uniform loads of CONSECUTIVE addresses (no clamp: the arrays are read up to
//           seven entries past the user's list, the row index is clamped instead) -> scalar loads, no v_readlane;
//         * the row address in SGPRs, the lane's byte offset as the load's 32-bit VGPR offset -> no 64-bit VALU arithmetic;
//         * unpack with the centring folded in: float bits = (24 bits << 6) + (centre << 23) by v_mad_u32_u24 (1 - 2 instructions
//           per value instead of 2; a zero entry becomes 2^(centre - 127), 2^-87 of the smallest term);
//         * packed fp32 math (v_pk_fma_f32, v_pk_mul_f32): two columns per instruction;
//         * RAW products of SUB (4 or 8) terms -- x is scaled by 2^centre so that a product of SUB terms stays a NORMAL fp32, whose
//           exponent field is then an exact integer sum and whose mantissa carries the same SUB - 1 roundings as the product of
//           mantissas -- and ONE frexp pair per sub-product; a sub-product that is not a positive normal number (a zero term:
//           quirk Q7; a term far outside the expected range) sends the wave through the per-term form for those rows;
//         * the mask of the user's own items before the loop (ballot over the list, one readlane per hit: ~1 hit per chunk).
//
// Both kernels are checked against a host fp64 evaluation of the same packed matrix on sampled (user, column) pairs, then timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/score_loop tools/micro/score_loop.hip && /tmp/score_loop [Ic] [users] [centre] [zeros]
// Run once, with the round's last GPU seconds (profiles/r4/micro_score_loop.txt: Ic 4096, 4096 users, 2.6e9 log terms, a 50 MB matrix):
//   v1 1.128 ms = 2.32e12 log terms/s (the library's kernels: 2.0 - 2.6e12), v2<4> 0.892 ms (1.26 x), v2<8> 0.916 ms (1.23 x); all checks
//   passed, v2's worst error against fp64 a third of v1's (1.7e-6 / 2.0e-6 against 6.9e-6 relative).  2.2 x fewer VALU instructions buy
//   1.25 x: behind the instruction count the loop meets the cache hierarchy's gather rate (8.7 TB/s of 768-byte row segments into the CUs).
// The STATIC count of the gfx950 code
// (`-S --cuda-device-only -o /tmp/score_loop.s`, hipcc of ROCm 7.2; VALU instructions on the hot path of one batch of eight rows =
// 32 log terms per lane):
//   v1      341  (42.6 per row -- the library's kernel measured 43.6 per row with the counters): 64 v_fmac, 32 v_perm + 32 v_lshrrev,
//                32 + 32 v_frexp_*, 36 v_add_u32, 24 v_readlane, 8 v_mad_u64_u32, 24 for the mask, 14 v_pk_mul, the fp64 tail
//   v2<8>   152  (19.0 per row): 32 v_pk_fma, 32 v_mad_u32_u24 + 16 v_perm + 8 v_lshrrev, 14 v_pk_mul, 8 v_lshl_add_u64 (the
//                compiler adds the SGPR row address to the lane offset in VALU instead of using the load's SGPR-base form), 4 v_cmp_class,
//                4 + 4 v_frexp_*, the same fp64 tail; idx / e / q arrive by three s_load_dwordx8 per batch
//   v2<4>   ~175 (22 per row): eight more v_frexp_*, four more multiplies / adds / class tests
// What made hipcc emit this: the read-only arrays as separate __restrict__ kernel arguments (else: vector loads + v_readfirstlane);
// unsigned 32 x 32 -> 64 row offsets behind readfirstlane (else: v_mad_u64_u32 per row); a global address-space pointer (else:
// flat_load); v_mad_u32_u24 by inline asm (else: shift + and + add); __builtin_amdgcn_classf (the generic builtin converts to fp64).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int P24_SHIFT = 6;   // float bits = packed << 6 (7 exponent + 17 mantissa bits of a float < 2)
constexpr int SB = 8;          // rows per batch
constexpr int CW = 256;        // columns per wave: four per lane
struct U3 {
    uint32_t a, b, c;
};
typedef float v2f __attribute__((ext_vector_type(2)));

struct Args {
    const unsigned char* __restrict__ M;   // [Ic rows][pitch bytes]: 24-bit entries
    long long pitch;
    int Ic;
    const float* __restrict__ a;       // l p_i per column
    const float* __restrict__ b;       // b_i per column
    const int* __restrict__ rowptr;    // [n_users + 1]
    const int* __restrict__ idx;       // rated items (row indices of M), ascending inside a user; padded by SB entries
    const float* __restrict__ e;       // e_uj per rating
    const float* __restrict__ q;       // q_j per rating
    const double* __restrict__ pv;     // per user
    float* __restrict__ S;             // [n_users][ldS]
    long long ldS;
    int n_users, n_slices;
    int centre;                        // v2: every term is scaled by 2^centre
};

__device__ __forceinline__ void unpack24(const U3& d, float* f) {
    f[0] = __uint_as_float(__builtin_amdgcn_perm(0u, d.a, 0x0201000cu) >> (8 - P24_SHIFT));
    f[1] = __uint_as_float(__builtin_amdgcn_perm(d.b, d.a, 0x0504030cu) >> (8 - P24_SHIFT));
    f[2] = __uint_as_float(__builtin_amdgcn_perm(d.c, d.b, 0x0403020cu) >> (8 - P24_SHIFT));
    f[3] = __uint_as_float(__builtin_amdgcn_perm(0u, d.c, 0x0302010cu) >> (8 - P24_SHIFT));
}

// ------------------------------------------------------------------------------------------------ v1: round 4's loop
__global__ __launch_bounds__(256) void k_v1(Args A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x / A.n_slices, slice = blockIdx.x - chunk * A.n_slices;
    const int col0 = chunk * CW, col = col0 + lane * 4;
    float a[4], bb[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        a[v] = col + v < A.Ic ? A.a[col + v] : 0.f;
        bb[v] = col + v < A.Ic ? A.b[col + v] : 0.f;
    }
    const char* __restrict__ Mcol = reinterpret_cast<const char*>(A.M) + (long long)col * 3;
    const double LN2 = 0.69314718055994530942;
    for (int u = slice * 4 + wave; u < A.n_users; u += A.n_slices * 4) {
        const int beg = A.rowptr[u], end = A.rowptr[u + 1];
        double t[4] = {0, 0, 0, 0};
        unsigned mask = 0;
        if (beg < end) {
            const int sub = lane & (SB - 1);
            int vi, vi_n;
            float ve, vq, ve_n, vq_n;
            {
                const int kk = min(beg + sub, end - 1);
                vi = A.idx[kk]; ve = A.e[kk]; vq = A.q[kk];
            }
            for (int k = beg; k < end; k += SB) {
                U3 g[SB];
                float e[SB], qq[SB];
                int jj[SB];
#pragma unroll
                for (int r = 0; r < SB; r++) {
                    jj[r] = __builtin_amdgcn_readlane(vi, r);
                    e[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ve), r));
                    qq[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vq), r));
                    g[r] = *reinterpret_cast<const U3*>(Mcol + (long long)jj[r] * A.pitch);
                }
                {
                    const int kk = min(k + SB + sub, end - 1);
                    vi_n = A.idx[kk]; ve_n = A.e[kk]; vq_n = A.q[kk];
                }
                float p[4] = {1.f, 1.f, 1.f, 1.f};
                int pe[4] = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < SB; r++) {
                    if (k + r < end) {
                        float gv[4];
                        unpack24(g[r], gv);
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const float x = fmaf(qq[r], bb[v], fmaf(a[v], e[r], gv[v]));
                            p[v] *= __builtin_amdgcn_frexp_mantf(x);
                            pe[v] += __builtin_amdgcn_frexp_expf(x);
                        }
                        const unsigned d = (unsigned)(jj[r] - col0);
                        if (d < (unsigned)CW && (int)(d / 4) == lane) mask |= 1u << (d % 4);
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; v++) t[v] += (double)__builtin_amdgcn_logf(p[v]) + (double)pe[v];
                vi = vi_n; ve = ve_n; vq = vq_n;
            }
        }
        float4 o;
        float* ov = reinterpret_cast<float*>(&o);
        const double base = A.pv[u];
#pragma unroll
        for (int v = 0; v < 4; v++) ov[v] = ((mask >> v) & 1u || col + v >= A.Ic) ? __builtin_nanf("") : (float)(base + LN2 * t[v]);
        *reinterpret_cast<float4*>(A.S + (long long)u * A.ldS + col) = o;
    }
}

// ------------------------------------------------------------------------------------------------ v2
// (the read-only arrays as separate `const T* __restrict__` kernel arguments, like score_body: only then are the wave-uniform loads
// proven invariant and issued as scalar loads)
__device__ __forceinline__ uint32_t mad24(uint32_t x, uint32_t cadd) {      // (x & 0xFFFFFF) * 64 + cadd in ONE instruction
    uint32_t r;                                                              // (hipcc turns __umul24(x, 64) + c into shift, and, add)
    asm("v_mad_u32_u24 %0, %1, 64, %2" : "=v"(r) : "v"(x), "s"(cadd));
    return r;
}
template <int SUB, bool CENTRED>
__global__ __launch_bounds__(256) void k_v2(const unsigned char* __restrict__ M_, const float* __restrict__ a_, const float* __restrict__ b_,
                                            const int* __restrict__ rowptr_, const int* __restrict__ idx_, const float* __restrict__ e_,
                                            const float* __restrict__ q_, const double* __restrict__ pv_, float* __restrict__ S_, Args A) {
    static_assert(SUB == 4 || SUB == 8, "rows per raw product");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x / A.n_slices, slice = blockIdx.x - chunk * A.n_slices;
    const int col0 = chunk * CW, col = col0 + lane * 4;
    // CENTRED (what profiles/r4/micro_score_loop.txt measured): every term scaled by 2^centre.  !CENTRED (what tools/patches/
    // score_body_v2.patch puts into the library): the terms as they are, the product of SUB = 4 of them STARTED at 2^108 -- an exact zero
    // term (every never co-rated pair at lambda = 0, quirk Q7) stays a zero and takes the term-by-term path to its -inf; with the centring
    // folded into the exponent field a zero entry became 2^(centre - 127) and could hide inside a normal product.
    static_assert(CENTRED || SUB == 4, "a start value covers four terms");
    const float sc = CENTRED ? __builtin_amdgcn_ldexpf(1.f, A.centre) : 1.f;
    const float P0 = CENTRED ? 1.f : 0x1p108f;
    v2f a01, a23, b01, b23;      // CENTRED: a and b scaled by 2^centre: x' = G' + a' e + q b'
    {
        float av[4], bv[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            av[v] = col + v < A.Ic ? a_[col + v] * sc : 0.f;
            bv[v] = col + v < A.Ic ? b_[col + v] * sc : 0.f;
        }
        a01 = v2f{av[0], av[1]}; a23 = v2f{av[2], av[3]};
        b01 = v2f{bv[0], bv[1]}; b23 = v2f{bv[2], bv[3]};
    }
    const uint32_t lane_off = (uint32_t)col * 3u;               // this lane's 12 bytes inside a row
    const uint32_t cadd = CENTRED ? (uint32_t)A.centre << 23 : 0u;   // 2^centre in the exponent field
    const int last_row = A.Ic - 1;
    const double LN2 = 0.69314718055994530942;
    for (int u = slice * 4 + wave; u < A.n_users; u += A.n_slices * 4) {
        const int beg = rowptr_[u], end = rowptr_[u + 1];
        // ---- the user's own items inside this chunk (the list is ascending: they are neighbours, ~1 per chunk)
        unsigned mask = 0;
        for (int k0 = beg; k0 < end; k0 += 64) {
            const int k = k0 + lane;
            const unsigned d = k < end ? (unsigned)(idx_[k] - col0) : 0xFFFFFFFFu;
            unsigned long long hit = __ballot(d < (unsigned)CW);
            while (hit) {
                const int l = __builtin_ctzll(hit);
                hit &= hit - 1;
                const unsigned dd = (unsigned)__builtin_amdgcn_readlane((int)d, l);
                if ((int)(dd >> 2) == lane) mask |= 1u << (dd & 3u);
            }
        }
        double t[4] = {0, 0, 0, 0};
        for (int k = beg; k < end; k += SB) {
            // wave-uniform, consecutive: scalar loads (the arrays are padded by SB entries; a row index read past the list is clamped)
            int jj[SB];
            float e[SB], qq[SB];
            U3 g[SB];
#pragma unroll
            for (int r = 0; r < SB; r++) {
                jj[r] = min(idx_[k + r], last_row);
                e[r] = e_[k + r];
                qq[r] = q_[k + r];
            }
#pragma unroll
            for (int r = 0; r < SB; r++) {
                // the row's address stays in an SGPR pair (s_mul_i32 / s_mul_hi_u32 / s_add_u32 / s_addc_u32), the lane's offset is the
                // load's 32-bit VGPR offset: global_load_dwordx3 v, v_off, s[rowp]  (left alone, hipcc moves jj into a VGPR and
                // multiplies with v_mad_u64_u32, a 64-bit VALU operation per row)
                const unsigned long long rb = (unsigned long long)M_ + (unsigned long long)(uint32_t)jj[r] * (unsigned long long)(uint32_t)A.pitch;
                const unsigned long long rs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb);
                typedef const __attribute__((address_space(1))) unsigned char* gptr;       // global, not flat
                typedef const __attribute__((address_space(1))) uint32_t* gword;
                const gword gw = reinterpret_cast<gword>(reinterpret_cast<gptr>(rs) + lane_off);
                g[r].a = gw[0]; g[r].b = gw[1]; g[r].c = gw[2];                            // one global_load_dwordx3
            }
            float m[4] = {1.f, 1.f, 1.f, 1.f};     // product of the sub-products' mantissas
            int ex[4] = {0, 0, 0, 0};              // sum of their exponents
            int n_folded = 0;                      // !CENTRED: folded sub-products, 2^108 each
#pragma unroll
            for (int r0 = 0; r0 < SB; r0 += SUB) {
                if (k + r0 >= end) break;
                v2f p01 = v2f{P0, P0}, p23 = v2f{P0, P0};
#pragma unroll
                for (int r = r0; r < r0 + SUB; r++) {
                    if (k + r < end) {
                        const U3 d = g[r];
                        v2f g01, g23;
                        g01.x = __uint_as_float(mad24(d.a, cadd));
                        g01.y = __uint_as_float(mad24(__builtin_amdgcn_alignbit(d.b, d.a, 24), cadd));
                        g23.x = __uint_as_float(mad24(__builtin_amdgcn_alignbit(d.c, d.b, 16), cadd));
                        g23.y = __uint_as_float(mad24(d.c >> 8, cadd));
                        const v2f e2 = v2f{e[r], e[r]}, q2 = v2f{qq[r], qq[r]};
                        p01 *= __builtin_elementwise_fma(q2, b01, __builtin_elementwise_fma(a01, e2, g01));
                        p23 *= __builtin_elementwise_fma(q2, b23, __builtin_elementwise_fma(a23, e2, g23));
                    }
                }
                const float pp[4] = {p01.x, p01.y, p23.x, p23.y};
                // a positive NORMAL product has an exact exponent and SUB - 1 roundings in its mantissa; anything else (zero, denormal,
                // inf, NaN) is done again term by term
                bool ok = true;
#pragma unroll
                for (int v = 0; v < 4; v++) ok = ok && __builtin_amdgcn_classf(pp[v], 0x100);      // +normal
                // (!CENTRED: none below 2^-90 either -- no partial product passed through the denormals, whatever the order of the terms)
                if (!CENTRED) ok = ok && fminf(fminf(pp[0], pp[1]), fminf(pp[2], pp[3])) > 0x1p-90f;
                if (__builtin_expect(__all(ok), 1)) {
                    n_folded++;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        m[v] *= __builtin_amdgcn_frexp_mantf(pp[v]);
                        ex[v] += __builtin_amdgcn_frexp_expf(pp[v]);
                    }
                } else {
#pragma unroll      // (static register indices: a rolled loop would put g[] into scratch)
                    for (int r = r0; r < r0 + SUB; r++) {
                        if (k + r >= end) continue;
                        float gv[4];
                        unpack24(g[r], gv);      // (a true zero stays zero here: log2 -> -inf like the reference's log(0))
                        const float av[4] = {a01.x, a01.y, a23.x, a23.y}, bv[4] = {b01.x, b01.y, b23.x, b23.y};
                        float mm[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const float x = fmaf(qq[r], bv[v], fmaf(av[v], e[r], gv[v] * sc));
                            mm[v] = __builtin_amdgcn_frexp_mantf(x);
                            ex[v] += __builtin_amdgcn_frexp_expf(x);
                        }
                        // (one term at a time into fp64: this path is rare and must not underflow the mantissa product either)
#pragma unroll
                        for (int v = 0; v < 4; v++) t[v] += (double)__builtin_amdgcn_logf(mm[v]);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 4; v++) t[v] += (double)__builtin_amdgcn_logf(m[v]) + (double)(ex[v] - (CENTRED ? 0 : 108 * n_folded));
        }
        float4 o;
        float* ov = reinterpret_cast<float*>(&o);
        const double base = pv_[u] - (CENTRED ? LN2 * (double)(end - beg) * (double)A.centre : 0.0);      // CENTRED: every term carried 2^centre
#pragma unroll
        for (int v = 0; v < 4; v++) ov[v] = ((mask >> v) & 1u || col + v >= A.Ic) ? __builtin_nanf("") : (float)(base + LN2 * t[v]);
        *reinterpret_cast<float4*>(S_ + (long long)u * A.ldS + col) = o;
    }
}

// ------------------------------------------------------------------------------------------------ host
static uint32_t pack24(float f) {      // 0 <= f < 2, round to nearest
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return (b + (1u << (P24_SHIFT - 1))) >> P24_SHIFT;
}
static float unpack24_host(uint32_t p) {
    const uint32_t b = p << P24_SHIFT;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

int main(int argc, char** argv) {
    const int Ic = argc > 1 ? atoi(argv[1]) : 8192;
    const int nU = argc > 2 ? atoi(argv[2]) : 4096;
    const int centre = argc > 3 ? atoi(argv[3]) : 20;
    // "zeros": a = q = 0 (lambda = 0), a term IS its matrix entry and 55 % of them are exact zeros: most scores are -inf.  The centred
    // variants cannot do this by construction (a zero entry becomes 2^(centre - 127)); v1 and the start-value variant must.
    const bool zeros = argc > 4 && std::strcmp(argv[4], "zeros") == 0;
    if (Ic < 256 || Ic % 4 || nU < 4) { fprintf(stderr, "Ic >= 256, a multiple of 4; users >= 4\n"); return 1; }
    const long long ldm = (Ic + 255) / 256 * 256, pitch = ldm * 3;
    std::mt19937_64 rng(20261005);
    auto unif = [&](double lo, double hi) { return lo + (hi - lo) * (double)(rng() >> 11) / 9007199254740992.0; };
    // matrix: 55 % zeros, the rest 2^-1 .. 2^-22 (stored scale: entries < 1), popular rows / columns larger
    std::vector<unsigned char> hM((size_t)Ic * pitch, 0);
    for (int j = 0; j < Ic; j++)
        for (int i = 0; i < Ic; i++) {
            const double pop = 1.0 + 12.0 * ((double)(j + i) / (2.0 * Ic));           // less popular -> smaller
            const float f = unif(0, 1) < 0.55 ? 0.f : (float)std::ldexp(unif(0.5, 1.0), -(int)(unif(1, 10) + pop));
            const uint32_t p = pack24(f);
            unsigned char* at = &hM[(size_t)j * pitch + (size_t)i * 3];
            at[0] = p & 255; at[1] = (p >> 8) & 255; at[2] = (p >> 16) & 255;
        }
    std::vector<float> ha(Ic), hb(Ic);
    for (int i = 0; i < Ic; i++) {
        ha[i] = zeros ? 0.f : (float)std::ldexp(unif(0.5, 1), -(int)unif(12, 30));       // l p_i
        hb[i] = (float)std::ldexp(unif(0.5, 1), -(int)unif(0, 10));        // b_i
    }
    std::vector<int> hrp(nU + 1, 0), hidx;
    std::vector<float> he, hq;
    for (int u = 0; u < nU; u++) {
        const int deg = u % 97 == 0 ? 0 : (int)std::min<double>(Ic / 2, std::exp(unif(std::log(20.0), std::log(600.0))));     // a few empty lists
        std::vector<int> row;
        while ((int)row.size() < deg) row.push_back((int)(std::pow(unif(0, 1), 2.0) * Ic) % Ic);      // popular items more often
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        for (int j : row) {
            hidx.push_back(j);
            he.push_back((float)std::ldexp(unif(0.5, 1), (int)unif(-8, 4)));       // e_uj
            hq.push_back(zeros ? 0.f : (float)std::ldexp(unif(0.5, 1), -(int)unif(14, 30)));     // q_j
        }
        hrp[u + 1] = (int)hidx.size();
    }
    const long long nnz = (long long)hidx.size();
    for (int r = 0; r < SB; r++) { hidx.push_back(0); he.push_back(1.f); hq.push_back(1.f); }      // the padding v2 may read
    std::vector<double> hpv(nU);
    for (int u = 0; u < nU; u++) hpv[u] = unif(100, 4000);
    const long long ldS = ldm;
    long long terms = 0;
    for (int u = 0; u < nU; u++) terms += (long long)(hrp[u + 1] - hrp[u]) * Ic;

    unsigned char* dM; float *da, *db, *de, *dq, *dS1, *dS2; int *drp, *didx; double* dpv;
    CHECK(hipMalloc(&dM, hM.size())); CHECK(hipMalloc(&da, Ic * 4)); CHECK(hipMalloc(&db, Ic * 4));
    CHECK(hipMalloc(&drp, (nU + 1) * 4)); CHECK(hipMalloc(&didx, hidx.size() * 4)); CHECK(hipMalloc(&de, he.size() * 4)); CHECK(hipMalloc(&dq, hq.size() * 4));
    CHECK(hipMalloc(&dpv, nU * 8)); CHECK(hipMalloc(&dS1, (size_t)nU * ldS * 4)); CHECK(hipMalloc(&dS2, (size_t)nU * ldS * 4));
    CHECK(hipMemcpy(dM, hM.data(), hM.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(da, ha.data(), Ic * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), Ic * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(drp, hrp.data(), (nU + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpv, hpv.data(), nU * 8, hipMemcpyHostToDevice));

    const int n_chunks = (int)(ldm / CW);
    const int n_slices = std::max(1, std::min((nU + 3) / 4, 4096 / n_chunks));
    Args A{dM, pitch, Ic, da, db, drp, didx, de, dq, dpv, dS1, ldS, nU, n_slices, centre};
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    auto run = [&](int which, float* S, const char* name) {
        A.S = S;
        CHECK(hipMemset(S, 0xFF, (size_t)nU * ldS * 4));
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(t0));
            if (which == 1) k_v1<<<n_chunks * n_slices, 256>>>(A);
            else if (which == 4) k_v2<4, true><<<n_chunks * n_slices, 256>>>(dM, da, db, drp, didx, de, dq, dpv, S, A);
            else if (which == 40) k_v2<4, false><<<n_chunks * n_slices, 256>>>(dM, da, db, drp, didx, de, dq, dpv, S, A);
            else k_v2<8, true><<<n_chunks * n_slices, 256>>>(dM, da, db, drp, didx, de, dq, dpv, S, A);
            CHECK(hipEventRecord(t1));
            CHECK(hipEventSynchronize(t1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) best = std::min(best, ms);
        }
        printf("%-28s %8.3f ms  %.3e log terms/s\n", name, best, (double)terms / (best * 1e-3));
        return best;
    };
    // ---- check: sampled (user, column) pairs against fp64 on the host
    auto check = [&](const float* dS, const char* name) {
        std::vector<float> hS((size_t)nU * ldS);
        CHECK(hipMemcpy(hS.data(), dS, hS.size() * 4, hipMemcpyDeviceToHost));
        std::mt19937_64 r2(7);
        double worst_rel = 0, worst_abs_logsum = 0;
        long long bad_mask = 0, checked = 0, n_inf = 0, wrong_inf = 0;
        for (int s = 0; s < 20000; s++) {
            const int u = (int)(r2() % (uint64_t)nU), i = (int)(r2() % (uint64_t)Ic);
            const int beg = hrp[u], end = hrp[u + 1];
            const bool rated = std::binary_search(hidx.begin() + beg, hidx.begin() + end, i);
            const float got = hS[(size_t)u * ldS + i];
            if (rated) { if (got == got) bad_mask++; continue; }
            double sum = 0;
            for (int k = beg; k < end; k++) {
                const unsigned char* at = &hM[(size_t)hidx[k] * pitch + (size_t)i * 3];
                const double gq = unpack24_host(at[0] | (at[1] << 8) | (at[2] << 16));
                sum += std::log2(gq + (double)ha[i] * (double)he[k] + (double)hq[k] * (double)hb[i]);
            }
            const double want = hpv[u] + 0.69314718055994530942 * sum;
            if (!(got == got)) { bad_mask++; continue; }
            if (std::isinf(want)) {      // a zero term: the reference's log(0)
                if (!(std::isinf(got) && got < 0)) wrong_inf++;
                n_inf++;
                continue;
            }
            worst_rel = std::max(worst_rel, std::fabs((double)got - (double)(float)want) / std::fabs(want));
            // the error of the log sum itself (what a nearly cancelling score sees), in units of ln: |got - want| with pv taken out
            worst_abs_logsum = std::max(worst_abs_logsum, std::fabs(((double)got - hpv[u]) - 0.69314718055994530942 * sum) - std::fabs(want) * 6e-8);
            checked++;
        }
        printf("%-28s %lld pairs: worst relative error of the score %.3e, worst |error of the log sum| beyond the float cast %.3e, wrong mask %lld, "
               "-inf scores %lld (wrong: %lld)\n", name, checked, worst_rel, std::max(0.0, worst_abs_logsum), bad_mask, n_inf, wrong_inf);
        return bad_mask == 0 && worst_rel < 1e-5 && wrong_inf == 0;
    };
    printf("Ic %d, users %d, ratings %lld, log terms %.3e, centre 2^%d, grid %d x 256 (%d chunks x %d slices)\n", Ic, nU, nnz, (double)terms, centre,
           n_chunks * n_slices, n_chunks, n_slices);
    bool ok = true;
    const float ms1 = run(1, dS1, "v1 (round 4's loop)");
    ok = check(dS1, "v1") && ok;
    float ms4 = ms1, ms8 = ms1;
    if (!zeros) {
        ms4 = run(4, dS2, "v2, products of 4 terms");
        ok = check(dS2, "v2<4>") && ok;
        ms8 = run(8, dS2, "v2, products of 8 terms");
        ok = check(dS2, "v2<8>") && ok;
    }
    const float ms40 = run(40, dS2, "v2, 4 terms from 2^108");
    ok = check(dS2, "v2<4, start value>") && ok;
    printf("speed-up over v1: %.2f (products of 4), %.2f (products of 8), %.2f (4 terms from 2^108: the library patch)   %s\n", ms1 / ms4, ms1 / ms8,
           ms1 / ms40, ok ? "CHECKS PASSED" : "CHECKS FAILED");
    return ok ? 0 : 2;
}
