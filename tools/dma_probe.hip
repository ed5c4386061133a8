// LDS-DMA address forms on gfx950: (a) global_load_lds_dwordx4 with a per-lane 64-bit pointer, (b) the same address formed as
// uniform base + 32-bit per-lane offset, (c) buffer_load_dwordx4 ... lds with a resource and a 32-bit offset; LDS targets above 64 KB.
//   hipcc -O3 --offload-arch=gfx950 -o tools/dma_probe tools/dma_probe.hip && tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FORM>
__global__ void __launch_bounds__(256) k(const double *src, double *dst, int pieces_per_wave, unsigned lds_off) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds_base = (unsigned) reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char *) smem) + lds_off;
    const char *base = reinterpret_cast<const char *>(src);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, 1 << 24, 0x00020000);
    for (int q = 0; q < pieces_per_wave; ++q) {
        const int t = wave + 4 * q;
        const unsigned off = (unsigned) (1024 * t + 16 * lane);
        const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + 1024u * (unsigned) t);
        if (FORM == 0) {
            const char *g = base + (unsigned long long) off;
            unsigned long long gg = (unsigned long long) g;
            asm volatile("" : "+v"(gg));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gg), "s"(l) : "memory");
        } else if (FORM == 1) {
            const char *g = base + off;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(l) : "memory");
        } else {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(off), "s"(rs), "s"(l) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const double *s = reinterpret_cast<const double *>(smem + lds_off);
    const int n = pieces_per_wave * 4 * 128;          // doubles staged
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = s[i];
}

int main() {
    const int ppw = 7, n = ppw * 4 * 128;
    std::vector<double> h(n), r(n);
    for (int i = 0; i < n; ++i) h[i] = 1.0 + i;
    double *src, *dst;
    CK(hipMalloc(&src, n * 8 + 4096)); CK(hipMalloc(&dst, n * 8));
    CK(hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice));
    const unsigned offs[3] = {0u, 60u * 1024u, 100u * 1024u};
    for (int form = 0; form < 3; ++form)
        for (unsigned lo : offs) {
            CK(hipMemset(dst, 0, n * 8));
            const size_t lds = lo + n * 8;
            if (form == 0) { CK(hipFuncSetAttribute((const void *) k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); k<0><<<1, 256, lds>>>(src, dst, ppw, lo); }
            if (form == 1) { CK(hipFuncSetAttribute((const void *) k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); k<1><<<1, 256, lds>>>(src, dst, ppw, lo); }
            if (form == 2) { CK(hipFuncSetAttribute((const void *) k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); k<2><<<1, 256, lds>>>(src, dst, ppw, lo); }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(r.data(), dst, n * 8, hipMemcpyDeviceToHost));
            int bad = 0, first = -1;
            for (int i = 0; i < n; ++i) if (r[i] != h[i]) { if (first < 0) first = i; ++bad; }
            printf("form %d (%s), LDS target +%u KB: %d of %d doubles wrong%s", form, form == 0 ? "global, per-lane 64-bit pointer" : form == 1 ? "global, uniform base + 32-bit offset" : "buffer ... lds", lo / 1024, bad, n, bad ? "" : "\n");
            if (bad) printf(" (first at %d: got %g want %g)\n", first, r[first], h[first]);
        }
    return 0;
}
