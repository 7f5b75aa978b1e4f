// dpp_check.hip -- confirms the DPP semantics the quad kernel relies on (gfx950):
//   update_dpp(old, src, row_ror:4K, row_mask 0xF, bank_mask M): lanes of bank b (lanes 4b..4b+3 of each
//   row of 16) with bit b of M set receive src of lane (i - 4K) mod 16 of their row; others keep old.
// Build: hipcc --offload-arch=gfx950 -O2 -o dpp_check dpp_check.hip ; prints OK or the first mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int K, int MASK>
__device__ int take(int old, int src)
{
    constexpr int ctrl = K == 0 ? 0xE4 : 0x120 + 4 * K;
    return __builtin_amdgcn_update_dpp(old, src, ctrl, 0xF, MASK, false);
}

__global__ void k(int *out)
{
    int lane = threadIdx.x;
    int src = 1000 + lane, old = -lane - 1;
    out[0 * 64 + lane] = take<0, 0x2>(old, src);
    out[1 * 64 + lane] = take<1, 0x4>(old, src);
    out[2 * 64 + lane] = take<2, 0x8>(old, src);
    out[3 * 64 + lane] = take<3, 0x1>(old, src);
    out[4 * 64 + lane] = take<1, 0xF>(old, src);
    out[5 * 64 + lane] = take<2, 0x5>(old, src);
}

int main()
{
    int *d, h[6 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int K[6] = {0, 1, 2, 3, 1, 2}, M[6] = {0x2, 0x4, 0x8, 0x1, 0xF, 0x5};
    int bad = 0;
    for (int t = 0; t < 6; t++)
        for (int lane = 0; lane < 64; lane++) {
            int row = lane & ~15, i = lane & 15, bank = i >> 2;
            int expect = (M[t] >> bank) & 1 ? 1000 + row + ((i - 4 * K[t]) & 15) : -lane - 1;
            if (h[t * 64 + lane] != expect) {
                if (bad < 10) printf("test %d lane %d: got %d expected %d\n", t, lane, h[t * 64 + lane], expect);
                bad++;
            }
        }
    printf(bad ? "MISMATCH %d\n" : "OK\n", bad);
    return bad != 0;
}
