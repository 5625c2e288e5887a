// Tile configurations of the bf16 GEMM that only tests and developer tools select (tools/gemm_bench.py, tools/gemm_race_screen.py,
// tests/test_gpu_parity.py: A/B references for the schedules the product path uses). Kept out of the hot-path translation unit.
#include "gemm_bf16_kernels.h"

int uv_gemm_diag_launch(const GemmArgs& a, int epilogue, int tile_cfg, hipStream_t s) {
    switch (tile_cfg) {
        case 2: return launch_cfg<256, 256, 2, 4>(a, epilogue, s);
        case 3: return launch_cfg<256, 128, 4, 2>(a, epilogue, s);
        case 4: return launch_cfg<256, 192, 2, 4>(a, epilogue, s);
        case 10: return launch_cfg<128, 128, 2, 2, 4>(a, epilogue, s);
        case 11: return launch_cfg<128, 128, 2, 4, 4>(a, epilogue, s);
        case 13: return launch_cfg<128, 128, 2, 4, 2>(a, epilogue, s);
        case 14: return launch_8ph<0>(a, epilogue, s);   // 4-phase schedule + fragment-wise read-modify-write epilogue (A/B reference)
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown tile_cfg %d", tile_cfg);
            return -1;
    }
}
