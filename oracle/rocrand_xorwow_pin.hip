/* TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing in the product may load this.
 *
 * An independent statement of the XORWOW recurrence to hold the oracle's (and, on the GPU, the product's) restatement against.
 * The reference draws every random number from cuRAND's XORWOW (utils/cuda_utility.cu:19-26 -> curand_uniform, rendering/rendering.cu:137
 * -> curand_init); cuRAND (CUDA toolkit, version unpinned by the reference's CMakeLists.txt:2) is absent from this image and so is
 * any header of it.  What IS in this image is AMD's rocRAND (ROCm 7.2.0, /opt/rocm/include/rocrand/rocrand_xorwow.h), whose
 * rocrand_device::xorwow_engine is a separate implementation of the same published generator (Marsaglia's xorwow: five xorshift words
 * + a Weyl sequence with increment 362437, output d + x[4]).  This harness of ours includes that header AS IT IS and exposes its
 * next() from caller-given state words, on the host and in a kernel.
 *
 * What this pins and what it does not:
 *   pinned      the state transition and the output word (orc_rng_next == rng_next of the product == xorwow_engine::next), from
 *               arbitrary state words, for as many steps as a test asks for;
 *   NOT pinned  curand_init's seed scramble -- rocRAND scrambles a seed with constants of its own (rocrand_xorwow.h:113-116:
 *               0x2c7f967f / 0xa03697cb / 1228688033 / 2073658381), so rocrand_init(seed) and curand_init(seed) start different
 *               streams by design; rr_rocrand_seed_words() exposes rocRAND's so that a test can state exactly that;
 *   NOT pinned  curand_uniform's float mapping -- rocRAND maps v to 2^-32 + v * 2^-32, cuRAND to v * 2^-32 + 2^-33
 *               (rocrand_uniform.h:65-68); rr_rocrand_uniform() is exposed for the same reason.
 * Both stay "restated from the published definition" in oracle/srt_oracle.c.
 */
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>
#include <stdint.h>

namespace {
// the engine keeps its words protected; a derived type may set them -- the header itself is untouched
struct engine_from_words : rocrand_device::xorwow_engine {
    __host__ __device__ engine_from_words(const uint32_t *w) : rocrand_device::xorwow_engine(0ull, 0ull, 0ull) {
        m_state.d = w[0];
        for (int i = 0; i < 5; i++) m_state.x[i] = w[1 + i];
    }
    __host__ __device__ void words(uint32_t *w) const {
        w[0] = m_state.d;
        for (int i = 0; i < 5; i++) w[1 + i] = m_state.x[i];
    }
};

// one lane per stream: n_steps draws from the stream's six words (d, x0..x4); the xor and the wrapping sum of the outputs, the last output
// and the words afterwards come back (64 bits of digest per stream + the complete end state: a wrong step cannot cancel out of both)
__global__ void rr_steps_kernel(const uint32_t *words_in, uint32_t n_streams, uint32_t n_steps, uint32_t *digest, uint32_t *words_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    uint32_t w[6];
    for (int k = 0; k < 6; k++) w[k] = words_in[(size_t)i * 6 + k];
    engine_from_words e(w);
    uint32_t x = 0, s = 0, last = 0;
    for (uint32_t k = 0; k < n_steps; k++) { last = e.next(); x ^= last; s += last * (2u * k + 1u); }
    digest[(size_t)i * 3 + 0] = x; digest[(size_t)i * 3 + 1] = s; digest[(size_t)i * 3 + 2] = last;
    e.words(w);
    for (int k = 0; k < 6; k++) words_out[(size_t)i * 6 + k] = w[k];
}
}  // namespace

#define RR_API extern "C" __attribute__((visibility("default")))

/* host: n draws from the six words (d, x0..x4); out[n] receives them (may be null), words_out[6] the state afterwards */
RR_API int rr_host_steps(const uint32_t *words_in, uint32_t n, uint32_t *out, uint32_t *words_out) {
    engine_from_words e(words_in);
    for (uint32_t k = 0; k < n; k++) { const uint32_t v = e.next(); if (out) out[k] = v; }
    e.words(words_out);
    return 0;
}

/* host: the same digest as the kernel's, for many streams (a CPU test compares the oracle's with it) */
RR_API int rr_host_digest(const uint32_t *words_in, uint32_t n_streams, uint32_t n_steps, uint32_t *digest, uint32_t *words_out) {
    for (uint32_t i = 0; i < n_streams; i++) {
        engine_from_words e(words_in + (size_t)i * 6);
        uint32_t x = 0, s = 0, last = 0;
        for (uint32_t k = 0; k < n_steps; k++) { last = e.next(); x ^= last; s += last * (2u * k + 1u); }
        digest[(size_t)i * 3 + 0] = x; digest[(size_t)i * 3 + 1] = s; digest[(size_t)i * 3 + 2] = last;
        e.words(words_out + (size_t)i * 6);
    }
    return 0;
}

/* rocRAND's OWN seeding and float mapping -- exposed so that a test can state that they are not cuRAND's (see the header) */
RR_API int rr_rocrand_seed_words(uint64_t seed, uint32_t *words_out) {
    struct peek : rocrand_device::xorwow_engine {
        __host__ peek(uint64_t s) : rocrand_device::xorwow_engine(s, 0ull, 0ull) {}
        __host__ void words(uint32_t *w) const { w[0] = m_state.d; for (int i = 0; i < 5; i++) w[1 + i] = m_state.x[i]; }
    } e(seed);
    e.words(words_out);
    return 0;
}
RR_API float rr_rocrand_uniform(uint32_t v) { return rocrand_device::detail::uniform_distribution(v); }

/* device: host pointers in, host pointers out (the harness owns its device memory); returns a hipError_t */
RR_API int rr_device_digest(const uint32_t *words_in, uint32_t n_streams, uint32_t n_steps, uint32_t *digest, uint32_t *words_out) {
    uint32_t *d_in = nullptr, *d_dig = nullptr, *d_out = nullptr;
    const size_t nb = (size_t)n_streams * 6 * sizeof(uint32_t), db = (size_t)n_streams * 3 * sizeof(uint32_t);
    hipError_t e;
    if ((e = hipMalloc(&d_in, nb)) != hipSuccess) return (int)e;
    if ((e = hipMalloc(&d_dig, db)) != hipSuccess) { (void)hipFree(d_in); return (int)e; }
    if ((e = hipMalloc(&d_out, nb)) != hipSuccess) { (void)hipFree(d_in); (void)hipFree(d_dig); return (int)e; }
    e = hipMemcpy(d_in, words_in, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rr_steps_kernel, dim3((n_streams + 255u) / 256u), dim3(256), 0, 0, d_in, n_streams, n_steps, d_dig, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(digest, d_dig, db, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(words_out, d_out, nb, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_dig); (void)hipFree(d_out);
    return (int)e;
}
