// layout check of buffer_load_dwordx4 ... lds on gfx950: where does lane L's 16 bytes land, with all lanes and with some disabled?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, float *dst, int nbytes, unsigned long long mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63;
    __attribute__((address_space(3))) char *lds = (__attribute__((address_space(3))) char *)smem;
    for (int i = lane; i < 512; i += 64) ((__attribute__((address_space(3))) f4v *)lds)[i] = f4v{-1.f, -1.f, -1.f, -1.f};
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    if ((mask >> lane) & 1ull) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds, 16, (63u - lane) * 16u, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds + 2048), 16, lane * 16u + 1024u, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    for (int i = lane; i < 512 * 4; i += 64) dst[i] = ((__attribute__((address_space(3))) float *)lds)[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = (float)i;
    float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 8192); hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (unsigned long long mask : {~0ull, 0x00000000ffff00ffull}) {
        k<<<1, 64, 8192>>>(s, d, 4096, mask);
        std::vector<float> o(2048); hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
        printf("mask %016llx\n first load  (lane L reads src chunk 63-L): lds chunk -> src chunk:", mask);
        for (int c = 0; c < 64; c++) printf(" %d", o[c * 4] < 0 ? -1 : (int)o[c * 4] / 4);
        printf("\n second load (lane L reads src chunk 64+L) at +2048:");
        for (int c = 0; c < 64; c++) printf(" %d", o[512 + c * 4] < 0 ? -1 : (int)o[512 + c * 4] / 4);
        printf("\n");
    }
    return 0;
}
