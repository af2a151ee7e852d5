// ref_sellmeier_driver.hip -- test harness around the ONE device function of the reference that compiles in this image without
// stand-ins: sellmeier_index (refraction/sellmeier.cu:11-22, compiled unmodified from /root/reference by oracle/Makefile's `ref`
// target with -fgpu-rdc and linked with this file).  It only declares the function as refraction/sellmeier.cuh does and calls it
// from a kernel, so that tests/test_ref_tables.py can compare the product's and the oracle's Sellmeier arithmetic with the
// reference's own compiled code on the GPU, bit for bit.  Test infrastructure; nothing here is part of the product.
#include <hip/hip_runtime.h>

__device__ float sellmeier_index(const float b[3], const float c[3], const float lambda);      // refraction/sellmeier.cuh:22-23

__global__ void ref_sellmeier_kernel(const float *b, const float *c, const float *lambda, unsigned n, float *out) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float bb[3] = {b[0], b[1], b[2]}, cc[3] = {c[0], c[1], c[2]};
    out[k] = sellmeier_index(bb, cc, lambda[k]);
}

// b, c: 3 floats each (host), lambda / out: n floats (host).  Returns 0 on success, a HIP error code otherwise.
extern "C" __attribute__((visibility("default"))) int ref_sellmeier_run(const float *b, const float *c, const float *lambda, unsigned n, float *out) {
    float *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, (size_t)(6 + 2 * (size_t)n) * sizeof(float));
    if (e != hipSuccess) return (int)e;
    float *db = d, *dc = d + 3, *dl = d + 6, *dout = d + 6 + n;
    e = hipMemcpy(db, b, 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dc, c, 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dl, lambda, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(ref_sellmeier_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, db, dc, dl, n, dout); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (int)e;
}
