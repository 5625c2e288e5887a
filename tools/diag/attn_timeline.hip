// Timeline of an attention launch (developer tool; not part of the library): compiles univid_amd/csrc/attention.hip with
// -DUV_ATTN_TIMELINE (100 MHz wall-clock stamps per workgroup at the phase boundaries of flash_attn_fwd3_kernel /
// flash_attn_fwd12_kernel + the CU it ran on) and prints where a workgroup slot's time goes - dispatch gap, prologue (Q rows + first
// K tile), key-tile loop, epilogue (O store) - per CU occupancy, the tail, and how the phases of different slots line up in time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DNDEBUG -DUV_ATTN_TIMELINE [-DUV_ATTN_TIMELINE_DRAIN] -I univid_amd/csrc \
//         tools/diag/attn_timeline.hip -o tools/diag/attn_timeline
//   tools/diag/attn_timeline [Lq 11440] [Lk 512] [batch 2]        (Lk >= 2048: the self-attention kernel)
#include "../../univid_amd/csrc/attention.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

void uv_set_error(const char* fmt, ...) { printf("uv_set_error: %s\n", fmt); }
int uv_launch_attn_pw4(const AttnArgs&, hipStream_t) { return -1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2b(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 11440, Lk = argc > 2 ? atoi(argv[2]) : 512, B = argc > 3 ? atoi(argv[3]) : 2;
    const int H = 24, D = 128, C = H * D;
    const int slots_per_cu = Lk >= 2048 ? 1 : 3;
    const long ldvt = (long)(B - 1) * Lk + (Lk + 63) / 64 * 64;
    std::vector<uint16_t> hq((size_t)B * L * C), hk((size_t)B * Lk * C), hv((size_t)C * ldvt);
    unsigned s = 12345;
    auto rnd = [&]() { float a = 0; for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (s >> 8) * (1.0f / 16777216.0f) - 0.5f; } return a * 1.732f; };
    for (auto& x : hq) x = f2b(rnd());
    for (auto& x : hk) x = f2b(rnd());
    for (auto& x : hv) x = f2b(rnd());
    uint16_t *q, *k, *vt, *out; unsigned long long* tl;
    int ids = Lk >= 2048 ? 4096 * 2 : (L + 127) / 128 * H * B;      // fwd12: an upper bound on its workgroups (unwritten rows stay zero)
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&vt, hv.size() * 2)); CK(hipMalloc(&out, hq.size() * 2));
    CK(hipMalloc(&tl, (size_t)ids * 8 * 8)); CK(hipMemset(tl, 0, (size_t)ids * 8 * 8));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(uv_attn_tl), &tl, sizeof(tl)));
    const float scale = 1.0f / sqrtf((float)D);
    auto launch = [&]() {
        if (uv_flash_attn_bf16(q, C, k, C, vt, ldvt, out, C, B, L, Lk, H, D, scale, nullptr)) { printf("launch failed\n"); exit(1); }
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)ids * 8);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));        // stamps of the LAST launch
    printf("Lq=%d Lk=%d B=%d: %d ids, %.1f us per launch (events, %d back-to-back)\n", L, Lk, B, ids, ms / reps * 1e3, reps);

    {   // compact: keep the ids that were stamped
        int n = 0;
        for (int i = 0; i < ids; ++i) if (h[i * 8]) { if (n != i) std::copy(h.begin() + i * 8, h.begin() + i * 8 + 8, h.begin() + n * 8); ++n; }
        printf("%d stamped workgroups\n", n);
        ids = n;
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < ids; ++i) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 3]); }
    printf("first entry -> last exit: %.1f us\n", (t1 - t0) * 0.01);
    // phases
    std::vector<double> pro, loop, epi, drain;
    for (int i = 0; i < ids; ++i) {
        pro.push_back((h[i * 8 + 1] - h[i * 8]) * 0.01); loop.push_back((h[i * 8 + 2] - h[i * 8 + 1]) * 0.01);
        epi.push_back((h[i * 8 + 3] - h[i * 8 + 2]) * 0.01);
        if (h[i * 8 + 4]) drain.push_back((h[i * 8 + 4] - h[i * 8 + 3]) * 0.01);
    }
    auto stat = [&](const char* n, std::vector<double> v) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= v.size();
        printf("  %-28s mean %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", n, m, v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    stat("prologue (Q rows, K tile 0)", pro); stat("key-tile loop", loop); stat("epilogue (O store issue)", epi); stat("store drain (vmcnt 0)", drain);
    // per CU: ids grouped by (xcc, se, cu); occupancy over time and gaps
    std::map<unsigned, std::vector<int>> cu;
    for (int i = 0; i < ids; ++i) {
        const unsigned hw = (unsigned)h[i * 8 + 7], xcc = (unsigned)(h[i * 8 + 7] >> 32) & 0xf;
        const unsigned key = (xcc << 16) | (hw & 0xff00);        // se_id[15:13] sh_id[12] cu_id[11:8]
        cu[key].push_back(i);
    }
    printf("%zu distinct CUs ran ids; ids per CU: ", cu.size());
    { std::vector<int> n; for (auto& kv : cu) n.push_back((int)kv.second.size()); std::sort(n.begin(), n.end());
      printf("min %d median %d max %d\n", n[0], n[n.size() / 2], n.back()); }
    // busy-slot integral per CU: sum of id durations / (3 x span)
    std::vector<double> occ, span, gap;
    for (auto& kv : cu) {
        auto& v = kv.second;
        unsigned long long a = ~0ull, b = 0; double busy = 0;
        for (int i : v) { a = std::min(a, h[i * 8]); b = std::max(b, h[i * 8 + 3]); busy += (h[i * 8 + 3] - h[i * 8]) * 0.01; }
        occ.push_back(busy / (slots_per_cu * (t1 - t0) * 0.01)); span.push_back((b - t0) * 0.01);
        // gaps: sort by entry; each entry (after the first three) is matched with the earliest unmatched exit before it
        std::vector<unsigned long long> ent, ex;
        for (int i : v) { ent.push_back(h[i * 8]); ex.push_back(h[i * 8 + 3]); }
        std::sort(ent.begin(), ent.end()); std::sort(ex.begin(), ex.end());
        for (size_t j = slots_per_cu; j < ent.size(); ++j) gap.push_back(((double)ent[j] - (double)ex[j - slots_per_cu]) * 0.01);
    }
    stat("slot occupancy per CU", occ); stat("CU's last exit (us from t0)", span); stat("exit -> next entry on the CU", gap);
    // how the phases line up: histogram over time of the number of ids in prologue / loop / epilogue, 2 us bins
    const double T = (t1 - t0) * 0.01; const double BW = T > 1000 ? 50.0 : 4.0; const int nbin = (int)(T / BW) + 1;
    std::vector<double> inpro(nbin, 0), inloop(nbin, 0), inepi(nbin, 0);
    auto add = [&](std::vector<double>& hist, unsigned long long a, unsigned long long b) {
        const double x = (a - t0) * 0.01, y = (b - t0) * 0.01;
        for (int bi = (int)(x / BW); bi <= (int)(y / BW) && bi < nbin; ++bi) {
            const double lo = std::max(x, bi * BW), hi = std::min(y, bi * BW + BW);
            if (hi > lo) hist[bi] += (hi - lo) / BW;
        }
    };
    for (int i = 0; i < ids; ++i) { add(inpro, h[i * 8], h[i * 8 + 1]); add(inloop, h[i * 8 + 1], h[i * 8 + 2]); add(inepi, h[i * 8 + 2], h[i * 8 + 3]); }
    printf("ids in flight by phase, %.0f us bins (prologue / loop / epilogue):\n", BW);
    for (int bi = 0; bi < nbin; ++bi) printf("  %5.0f us  %6.1f %6.1f %6.1f\n", bi * BW, inpro[bi], inloop[bi], inepi[bi]);
    return 0;
}
