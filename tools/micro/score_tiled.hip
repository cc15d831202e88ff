// Does the plain full pass of a MANY-CLUSTER job get its matrix rows out of the L2 when the users of a column chunk sweep the rows
// together?    Evidence it answers to: profiles/r4/k50_n1000_pmc_*.txt -- at 50 clusters x N = 1000
// `k_score` moves 3.59 TB per job through the fabric (2.8 B per log term, L2 hit rate 13 %, 8.5 TB/s while it runs): a 3 250-user
// cluster reads every 768-byte row segment of a 256-column chunk 17 times, but each wave walks ITS user's list from the first row to
// the last, the waves of a chunk are spread over all eight XCDs, and a chunk's 25 MB of rows pass through eight 4 MB L2s in no order.
//
// One synthetic cluster (default: 32 768 items -> a 3.2 GB 24-bit matrix generated on the device, 3 250 users, ~150 ratings each):
//   flat   today's arrangement: workgroup = (chunk, slice of the users), blockIdx round-robin over the XCDs, a wave walks one user's
//          whole list, then the next user's;
//   tiled  the same row loop, but (a) all workgroups of a chunk sit on ONE XCD (blockIdx % 8 == chunk % 8), (b) a wave owns UPW users
//          for the whole chunk and walks ROW BLOCKS of RB rows outermost -- block 0 of all its users, block 1, ... -- with the partial
//          log sums of its users in registers (fp64, 8 VGPRs per user), so that at any time the waves of a chunk are inside the same
//          few 768 B x RB tiles (RB = 4096: 3 MB).  No barrier between the waves: the order is a tendency, not a guarantee -- which is
//          exactly what has to be measured.  The list of a user is ascending, a row block is a sub-range of it (`off` table).
// Both use the rewritten row loop of tools/micro/score_loop.hip (products of four terms started at 2^108).  Checked against each other
// (all scores) and against fp64 on the host for sampled pairs (the host regenerates matrix entries from the same hash).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/score_tiled tools/micro/score_tiled.hip && /tmp/score_tiled [Ic] [users] [RB]
//   rocprofv3 --pmc FETCH_SIZE -- /tmp/score_tiled   (and TCC_HIT_sum TCC_MISS_sum) shows what the two arrangements fetch.
// NOT YET RUN (written after the round's GPU minutes were spent); compiles for gfx950.
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

constexpr int P24_SHIFT = 6;
constexpr int SB = 8;
constexpr int CW = 256;
constexpr int UPW = 2;      // users a wave of the tiled kernel owns
struct U3 {
    uint32_t a, b, c;
};
typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- the matrix: a hash of (row, column), the same on host and device
__host__ __device__ inline uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
// packed 24-bit entry of (row j, column i): 55 % zeros, else a value in 2^-1 .. 2^-22 (popular = low index = larger)
__host__ __device__ inline uint32_t entry24(uint32_t j, uint32_t i, uint32_t Ic) {
    const uint32_t h = mix32(((uint64_t)j << 32) | i);
    if ((h & 1023u) < 563u) return 0u;
    const uint32_t pop = 1u + (uint32_t)((12ull * ((uint64_t)j + i)) / (2ull * Ic));        // (integers: the same on host and device)
    const uint32_t ex = 127u - (1u + ((h >> 10) % 10u) + pop);                  // exponent field
    const uint32_t mant = (h >> 14) & 0x1FFFFu;                                  // 17 mantissa bits
    return (ex << 17) | mant;                                                    // = float bits >> 6
}
__global__ void k_fill(unsigned char* __restrict__ M, long long pitch, int Ic) {
    const long long n4 = (long long)Ic * (Ic / 4);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long long)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(t / (Ic / 4)), i4 = (uint32_t)(t % (Ic / 4)) * 4u;
        const uint32_t p0 = entry24(j, i4, Ic), p1 = entry24(j, i4 + 1, Ic), p2 = entry24(j, i4 + 2, Ic), p3 = entry24(j, i4 + 3, Ic);
        U3 o{p0 | (p1 << 24), (p1 >> 8) | (p2 << 16), (p2 >> 16) | (p3 << 8)};
        *reinterpret_cast<U3*>(M + (long long)j * pitch + (long long)i4 * 3) = o;
    }
}

// ---------------------------------------------------------------- the row loop (tools/micro/score_loop.hip, k_v2<4, false>)
__device__ __forceinline__ float p24_bits(uint32_t x) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, 64" : "=v"(r) : "v"(x));
    return __uint_as_float(r);
}
__device__ __forceinline__ void unpack24(const U3& d, float* f) {
    f[0] = __uint_as_float(__builtin_amdgcn_perm(0u, d.a, 0x0201000cu) >> (8 - P24_SHIFT));
    f[1] = __uint_as_float(__builtin_amdgcn_perm(d.b, d.a, 0x0504030cu) >> (8 - P24_SHIFT));
    f[2] = __uint_as_float(__builtin_amdgcn_perm(d.c, d.b, 0x0403020cu) >> (8 - P24_SHIFT));
    f[3] = __uint_as_float(__builtin_amdgcn_perm(0u, d.c, 0x0302010cu) >> (8 - P24_SHIFT));
}
// rows [beg, end) of one user's list against this lane's four columns: t[v] += log2 sums
__device__ __forceinline__ void walk_rows(const unsigned char* __restrict__ M_, long long pitch, uint32_t lane_off, const int* __restrict__ idx_,
                                          const float* __restrict__ e_, const float* __restrict__ q_, int beg, int end, v2f a01, v2f a23, v2f b01,
                                          v2f b23, double* t) {
    if (beg >= end) return;
    const float P0 = 0x1p108f;
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    typedef const __attribute__((address_space(1))) uint32_t* gword;
    auto batch = [&](int k, auto tail_tag) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        int jj[SB];
        float e[SB], qq[SB];
        U3 g[SB];
#pragma unroll
        for (int r = 0; r < SB; r++) {
            const int kk = TAIL ? min(k + r, end - 1) : k + r;
            jj[r] = idx_[kk]; e[r] = e_[kk]; qq[r] = q_[kk];
        }
#pragma unroll
        for (int r = 0; r < SB; r++) {
            const unsigned long long rb = (unsigned long long)M_ + (unsigned long long)(uint32_t)jj[r] * (unsigned long long)pitch;
            const unsigned long long rs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb);
            const gword gw = reinterpret_cast<gword>(reinterpret_cast<gptr>(rs) + lane_off);
            g[r].a = gw[0]; g[r].b = gw[1]; g[r].c = gw[2];
        }
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        int ex[4] = {0, 0, 0, 0};
        int n_folded = 0;
#pragma unroll
        for (int r0 = 0; r0 < SB; r0 += 4) {
            if (TAIL && k + r0 >= end) break;
            v2f p01 = v2f{P0, P0}, p23 = v2f{P0, P0};
#pragma unroll
            for (int r = r0; r < r0 + 4; r++) {
                if (!TAIL || k + r < end) {
                    const U3 d = g[r];
                    v2f g01, g23;
                    g01.x = p24_bits(d.a);
                    g01.y = p24_bits(__builtin_amdgcn_alignbit(d.b, d.a, 24));
                    g23.x = p24_bits(__builtin_amdgcn_alignbit(d.c, d.b, 16));
                    g23.y = p24_bits(d.c >> 8);
                    const v2f e2 = v2f{e[r], e[r]}, q2 = v2f{qq[r], qq[r]};
                    p01 *= __builtin_elementwise_fma(q2, b01, __builtin_elementwise_fma(a01, e2, g01));
                    p23 *= __builtin_elementwise_fma(q2, b23, __builtin_elementwise_fma(a23, e2, g23));
                }
            }
            const float pp[4] = {p01.x, p01.y, p23.x, p23.y};
            bool ok = fminf(fminf(pp[0], pp[1]), fminf(pp[2], pp[3])) > 0x1p-90f;
#pragma unroll
            for (int v = 0; v < 4; v++) ok = ok && __builtin_amdgcn_classf(pp[v], 0x100);
            if (__builtin_expect(__all(ok), 1)) {
                n_folded++;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    m[v] *= __builtin_amdgcn_frexp_mantf(pp[v]);
                    ex[v] += __builtin_amdgcn_frexp_expf(pp[v]);
                }
            } else {
#pragma unroll
                for (int r = r0; r < r0 + 4; r++) {
                    if (TAIL && k + r >= end) continue;
                    float gv[4];
                    unpack24(g[r], gv);
                    const float av[4] = {a01.x, a01.y, a23.x, a23.y}, bv[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const float x = fmaf(qq[r], bv[v], fmaf(av[v], e[r], gv[v]));
                        t[v] += (double)__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(x));
                        ex[v] += __builtin_amdgcn_frexp_expf(x);
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; v++) t[v] += (double)__builtin_amdgcn_logf(m[v]) + (double)(ex[v] - 108 * n_folded);
    };
    int k = beg;
    for (; k + SB <= end; k += SB) batch(k, std::false_type{});
    if (k < end) batch(k, std::true_type{});
}

struct Args {
    long long pitch, ldS;
    int Ic, n_users, n_slices, n_rb, rb_rows;
};

__device__ __forceinline__ void load_ab(const float* __restrict__ a_, const float* __restrict__ b_, int col, int Ic, v2f& a01, v2f& a23, v2f& b01, v2f& b23) {
    float av[4], bv[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        av[v] = col + v < Ic ? a_[col + v] : 0.f;
        bv[v] = col + v < Ic ? b_[col + v] : 0.f;
    }
    a01 = v2f{av[0], av[1]}; a23 = v2f{av[2], av[3]};
    b01 = v2f{bv[0], bv[1]}; b23 = v2f{bv[2], bv[3]};
}
__device__ __forceinline__ void store4(float* __restrict__ S_, long long ldS, int u, int col, int Ic, double base, const double* t) {
    const double LN2 = 0.69314718055994530942;
    float4 o;
    float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int v = 0; v < 4; v++) ov[v] = col + v >= Ic ? __builtin_nanf("") : (float)(base + LN2 * t[v]);
    *reinterpret_cast<float4*>(S_ + (long long)u * ldS + col) = o;
}

// flat: workgroup = (chunk, slice); a wave walks whole lists, user after user  (the "already rated" mask is left out of both kernels)
__global__ __launch_bounds__(256) void k_flat(const unsigned char* __restrict__ M_, const float* __restrict__ a_, const float* __restrict__ b_,
                                              const int* __restrict__ rowptr_, const int* __restrict__ idx_, const float* __restrict__ e_,
                                              const float* __restrict__ q_, const double* __restrict__ pv_, float* __restrict__ S_, Args A) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x / A.n_slices, slice = blockIdx.x - chunk * A.n_slices;
    const int col = chunk * CW + lane * 4;
    v2f a01, a23, b01, b23;
    load_ab(a_, b_, col, A.Ic, a01, a23, b01, b23);
    for (int u = slice * 4 + wave; u < A.n_users; u += A.n_slices * 4) {
        double t[4] = {0, 0, 0, 0};
        walk_rows(M_, A.pitch, (uint32_t)col * 3u, idx_, e_, q_, rowptr_[u], rowptr_[u + 1], a01, a23, b01, b23, t);
        store4(S_, A.ldS, u, col, A.Ic, pv_[u], t);
    }
}

// tiled: all workgroups of a chunk on one XCD, a wave owns UPW users and sweeps the row blocks outermost
__global__ __launch_bounds__(256) void k_tiled(const unsigned char* __restrict__ M_, const float* __restrict__ a_, const float* __restrict__ b_,
                                               const int* __restrict__ off_ /* [n_users][n_rb + 1] */, const int* __restrict__ idx_,
                                               const float* __restrict__ e_, const float* __restrict__ q_, const double* __restrict__ pv_,
                                               float* __restrict__ S_, Args A, int n_chunks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx = ((chunk / 8) * n_slices + slice) * 8 + chunk % 8: consecutive workgroup ids go round the eight XCDs, so chunk c stays on XCD c % 8
    const int x = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int slice = rest % A.n_slices, chunk = (rest / A.n_slices) * 8 + x;
    if (chunk >= n_chunks) return;
    const int col = chunk * CW + lane * 4;
    v2f a01, a23, b01, b23;
    load_ab(a_, b_, col, A.Ic, a01, a23, b01, b23);
    const int u0 = (slice * 4 + wave) * UPW;
    if (u0 >= A.n_users) return;
    double t[UPW][4];
#pragma unroll
    for (int w = 0; w < UPW; w++)
#pragma unroll
        for (int v = 0; v < 4; v++) t[w][v] = 0.0;
    for (int rb = 0; rb < A.n_rb; rb++) {
#pragma unroll
        for (int w = 0; w < UPW; w++) {
            const int u = u0 + w;
            if (u < A.n_users) {      // (wave-uniform)
                const int* __restrict__ o = off_ + (long long)u * (A.n_rb + 1) + rb;
                walk_rows(M_, A.pitch, (uint32_t)col * 3u, idx_, e_, q_, o[0], o[1], a01, a23, b01, b23, t[w]);
            }
        }
    }
#pragma unroll
    for (int w = 0; w < UPW; w++)
        if (u0 + w < A.n_users) store4(S_, A.ldS, u0 + w, col, A.Ic, pv_[u0 + w], t[w]);
}

static float unpack24_host(uint32_t p) {
    const uint32_t b = p << P24_SHIFT;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

int main(int argc, char** argv) {
    const int Ic = argc > 1 ? atoi(argv[1]) : 32768;
    const int nU = argc > 2 ? atoi(argv[2]) : 3250;
    const int RB = argc > 3 ? atoi(argv[3]) : 4096;
    if (Ic < 256 || Ic % 256 || nU < 8 || RB < 256) { fprintf(stderr, "Ic a multiple of 256, users >= 8, RB >= 256\n"); return 1; }
    const long long ldm = Ic, pitch = ldm * 3, ldS = ldm;
    const int n_rb = (Ic + RB - 1) / RB, n_chunks = Ic / CW;
    std::mt19937_64 rng(20261006);
    auto unif = [&](double lo, double hi) { return lo + (hi - lo) * (double)(rng() >> 11) / 9007199254740992.0; };
    std::vector<float> ha(Ic), hb(Ic);
    for (int i = 0; i < Ic; i++) {
        ha[i] = (float)std::ldexp(unif(0.5, 1), -(int)unif(12, 30));
        hb[i] = (float)std::ldexp(unif(0.5, 1), -(int)unif(0, 10));
    }
    std::vector<int> hrp(nU + 1, 0), hidx, hoff((size_t)nU * (n_rb + 1));
    std::vector<float> he, hq;
    for (int u = 0; u < nU; u++) {
        const int deg = (int)std::min<double>(Ic / 4, std::exp(unif(std::log(20.0), std::log(1100.0))));       // mean ~ 150 - 270
        std::vector<int> row;
        while ((int)row.size() < deg) row.push_back((int)(std::pow(unif(0, 1), 2.0) * Ic) % Ic);
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        for (int j : row) {
            hidx.push_back(j);
            he.push_back((float)std::ldexp(unif(0.5, 1), (int)unif(-8, 4)));
            hq.push_back((float)std::ldexp(unif(0.5, 1), -(int)unif(14, 30)));
        }
        hrp[u + 1] = (int)hidx.size();
        for (int rb = 0; rb <= n_rb; rb++)       // first position of the list with a row >= rb * RB
            hoff[(size_t)u * (n_rb + 1) + rb] =
                (int)(std::lower_bound(hidx.begin() + hrp[u], hidx.begin() + hrp[u + 1], (long long)rb * RB > Ic ? Ic : rb * RB) - hidx.begin());
    }
    const long long nnz = (long long)hidx.size();
    std::vector<double> hpv(nU);
    for (int u = 0; u < nU; u++) hpv[u] = unif(100, 4000);
    const double terms = (double)nnz * Ic;

    unsigned char* dM; float *da, *db, *de, *dq, *dS1, *dS2; int *drp, *didx, *doff; double* dpv;
    CHECK(hipMalloc(&dM, (size_t)Ic * pitch)); CHECK(hipMalloc(&da, Ic * 4)); CHECK(hipMalloc(&db, Ic * 4));
    CHECK(hipMalloc(&drp, (nU + 1) * 4)); CHECK(hipMalloc(&didx, hidx.size() * 4)); CHECK(hipMalloc(&de, he.size() * 4)); CHECK(hipMalloc(&dq, hq.size() * 4));
    CHECK(hipMalloc(&doff, hoff.size() * 4)); CHECK(hipMalloc(&dpv, nU * 8));
    CHECK(hipMalloc(&dS1, (size_t)nU * ldS * 4)); CHECK(hipMalloc(&dS2, (size_t)nU * ldS * 4));
    CHECK(hipMemcpy(da, ha.data(), Ic * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), Ic * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(drp, hrp.data(), (nU + 1) * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(doff, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpv, hpv.data(), nU * 8, hipMemcpyHostToDevice));
    k_fill<<<4096, 256>>>(dM, pitch, Ic);
    CHECK(hipDeviceSynchronize());

    Args A{pitch, ldS, Ic, nU, 0, n_rb, RB};
    const int slices_flat = std::max(1, std::min((nU + 3) / 4, 65536 / n_chunks));
    const int slices_tiled = (nU + 4 * UPW - 1) / (4 * UPW);
    const int chunks8 = (n_chunks + 7) / 8 * 8;
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    auto run = [&](bool tiled, float* S, const char* name) {
        CHECK(hipMemset(S, 0xFF, (size_t)nU * ldS * 4));
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(t0));
            if (tiled) { A.n_slices = slices_tiled; k_tiled<<<chunks8 * slices_tiled, 256>>>(dM, da, db, doff, didx, de, dq, dpv, S, A, n_chunks); }
            else { A.n_slices = slices_flat; k_flat<<<n_chunks * slices_flat, 256>>>(dM, da, db, drp, didx, de, dq, dpv, S, A); }
            CHECK(hipEventRecord(t1));
            CHECK(hipEventSynchronize(t1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) best = std::min(best, ms);
        }
        printf("%-46s %9.3f ms  %.3e log terms/s  (3 B per term from anywhere: %.2f TB/s)\n", name, best, terms / (best * 1e-3), 3.0 * terms / (best * 1e-3) / 1e12);
        return best;
    };
    printf("Ic %d (matrix %.2f GB), users %d, ratings %lld (%.0f per user), log terms %.3e, row blocks of %d rows (%d), tile %.2f MB\n", Ic,
           (double)Ic * pitch / 1e9, nU, nnz, (double)nnz / nU, terms, RB, n_rb, RB * 768.0 / 1e6);
    const float ms_flat = run(false, dS1, "flat (a wave walks whole lists)");
    const float ms_tiled = run(true, dS2, "tiled (one XCD per chunk, row blocks outermost)");
    // ---- every score of the two arrangements against each other, sampled pairs against fp64
    std::vector<float> h1((size_t)nU * ldS), h2((size_t)nU * ldS);
    CHECK(hipMemcpy(h1.data(), dS1, h1.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), dS2, h2.size() * 4, hipMemcpyDeviceToHost));
    double worst_pair = 0;
    long long nan_mismatch = 0;
    for (size_t k = 0; k < h1.size(); k++) {
        const float x = h1[k], y = h2[k];
        if ((x == x) != (y == y)) { nan_mismatch++; continue; }
        if (x == x) worst_pair = std::max(worst_pair, std::fabs((double)x - (double)y) / std::fabs((double)x));
    }
    std::mt19937_64 r2(7);
    double worst_ref = 0;
    for (int s = 0; s < 4000; s++) {
        const int u = (int)(r2() % (uint64_t)nU), i = (int)(r2() % (uint64_t)Ic);
        double sum = 0;
        for (int k = hrp[u]; k < hrp[u + 1]; k++)
            sum += std::log2((double)unpack24_host(entry24((uint32_t)hidx[k], (uint32_t)i, (uint32_t)Ic)) + (double)ha[i] * (double)he[k] + (double)hq[k] * (double)hb[i]);
        const double want = hpv[u] + 0.69314718055994530942 * sum;
        worst_ref = std::max(worst_ref, std::fabs((double)h2[(size_t)u * ldS + i] - (double)(float)want) / std::fabs(want));
    }
    // (the two arrangements add the same batches in different groupings -- a row block starts a new batch --: equal up to fp32 / fp64 rounding)
    const bool ok = nan_mismatch == 0 && worst_pair < 2e-6 && worst_ref < 1e-5;
    printf("tiled against flat on all %zu scores: worst relative difference %.3e; tiled against fp64 on 4000 pairs: %.3e   %s\n", h1.size(), worst_pair,
           worst_ref, ok ? "CHECKS PASSED" : "CHECKS FAILED");
    printf("tiled / flat: %.2f x\n", ms_flat / ms_tiled);
    return ok ? 0 : 2;
}
