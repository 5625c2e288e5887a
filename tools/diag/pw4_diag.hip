// Diagnostic build of the 4-wave x 64-query self-attention kernel (developer tool; not part of the library): compiles
// tools/diag/attn_pw4.hip with -DUV_PW4_DIAG (in-kernel s_memtime / s_memrealtime stamps around the main loop, timing-only
// ablations -DUV_PW4_ABL=<bits>) and runs it on random data at the DiT's self-attention shape. Prints wall time per launch,
// stamped cycles per key-tile iteration (64 MFMAs per wave) and the clock the chip held in the loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DUV_PW4_DIAG [-DUV_PW4_ABL=n] -I univid_amd/csrc \
//         tools/diag/pw4_diag.hip -o tools/diag/pw4_diag_<n>
#include "attn_pw4.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

void uv_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2b(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 11440, B = argc > 2 ? atoi(argv[2]) : 2, H = 24, D = 128, C = H * D;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const long ldvt = (long)(B - 1) * L + (L + 63) / 64 * 64;
    std::vector<uint16_t> hq((size_t)B * L * C), hk((size_t)B * L * C), hv((size_t)C * ldvt);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; // sum of 4 uniforms ~ gaussian-ish, unit variance
        float a = 0; for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (s >> 8) * (1.0f / 16777216.0f) - 0.5f; } return a * 1.732f; };
    for (auto& x : hq) x = f2b(rnd());
    for (auto& x : hk) x = f2b(rnd());
    for (auto& x : hv) x = f2b(rnd());
    uint16_t *q, *k, *vt, *out; unsigned long long* dbg;
    const int q_blocks = (L + 255) / 256, nwg = q_blocks * H * B;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&vt, hv.size() * 2)); CK(hipMalloc(&out, hq.size() * 2));
    CK(hipMalloc(&dbg, (size_t)nwg * 4 * 4 * 8)); CK(hipMemset(dbg, 0, (size_t)nwg * 4 * 4 * 8));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(uv_pw4_dbg), &dbg, sizeof(dbg)));
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.out = out; a.ldq = a.ldk = a.ldo = C; a.ldvt = ldvt; a.Lq = a.Lk = L; a.H = H; a.batch = B; a.n12 = 0;
    a.q_blocks = q_blocks; a.scale_log2 = 1.4426950408889634f / sqrtf((float)D);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) uv_launch_attn_pw4(a, 0);
    CK(hipDeviceSynchronize());
    // >= 2 s of back-to-back launches before the stamped one (the clock the chip HOLDS, MI355X_MICROARCH.md DVFS item 6)
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) uv_launch_attn_pw4(a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hd((size_t)nwg * 16);
    CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int w = 0; w < nwg * 4; ++w) {
        const unsigned long long c = hd[w * 4], r = hd[w * 4 + 1], it = hd[w * 4 + 2];
        if (it > 0 && r > 0) { cyc.push_back((double)c / it); clk.push_back((double)c / r * 0.1); }   // s_memrealtime ticks at 100 MHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double fl = 4.0 * B * L * (double)L * C;
    printf("ABL=%d L=%d B=%d: %.3f ms/launch  %.1f TFLOP/s | main loop: median %.0f cycles per tile iteration (min %.0f, p90 %.0f) = %.1f cycles per MFMA; clock in the loop %.2f GHz (median)\n",
           UV_PW4_ABL, L, B, ms / reps, fl / (ms / reps) / 1e9, cyc[cyc.size() / 2], cyc[0], cyc[cyc.size() * 9 / 10], cyc[cyc.size() / 2] / 64.0,
           clk[clk.size() / 2]);
    return 0;
}
