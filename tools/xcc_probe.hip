// Where do the workgroups of a launch land?  Records HW_REG_XCC_ID per workgroup for several grids / repeated launches and
// reports whether "blockIdx % 8 == one XCD" holds (the placement assumption behind XCD-local exchanges).
//   hipcc -O3 --offload-arch=gfx950 tools/xcc_probe.hip -o tools/bin/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void rec(int* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;      // HW_REG_XCC_ID[3:0]
        out[blockIdx.x] = (int)x;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);                  // keep the workgroup resident a while
}

int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    std::vector<int> h(4096);
    for (int grid : {64, 128, 256, 320, 512}) {
        int bad_launches = 0, launches = 200;
        for (int l = 0; l < launches; ++l) {
            rec<<<grid, 512>>>(d, (l % 3) * 50);
            hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
            int first[8]; for (int i = 0; i < 8; ++i) first[i] = h[i];
            bool ok = true;
            for (int b = 0; b < grid; ++b) ok = ok && h[b] == first[b % 8];
            for (int i = 0; i < 8; ++i) for (int j = 0; j < i; ++j) ok = ok && first[i] != first[j];
            bad_launches += !ok;
            if (l == 0) { printf("grid %d first wgs:", grid); for (int b = 0; b < 16 && b < grid; ++b) printf(" %d", h[b]); printf("\n"); }
        }
        printf("grid %4d: %d of %d launches broke the blockIdx %% 8 <-> XCD map\n", grid, bad_launches, launches);
    }
    return 0;
}
