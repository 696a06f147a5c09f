// Probe: what does `buffer_load_dwordx4 ... offen lds` write to LDS for lanes whose offset is out of the descriptor's range,
// and is the SGPR offset part of the range check?  (hipcc --offload-arch=gfx950 -O3 tools/probes/buffer_lds_oob.hip -o <exe>)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned char* g, int nrec, int soff, int mode, unsigned* out) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 1024 / 4; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nrec, 0x00020000);
    int voff = threadIdx.x * 16;
    if (mode == 1 && (threadIdx.x & 1)) voff = 0x7fffff00;     // odd lanes far out of range
    if (mode == 2 && (threadIdx.x & 1)) voff = nrec;           // odd lanes just out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    const int N = 4096;
    std::vector<unsigned> h(N / 4);
    for (int i = 0; i < N / 4; ++i) h[i] = 0x1000 + i;
    unsigned char* d; unsigned* o;
    hipMalloc(&d, N); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    std::vector<unsigned> r(256);
    struct { int nrec, soff, mode; const char* what; } cases[] = {
        {N, 0, 0, "all in range"}, {N, 0, 1, "odd lanes voffset 0x7fffff00"}, {N, 0, 2, "odd lanes voffset = num_records"},
        {1024, 0, 0, "num_records 1024 = exactly the 64 lanes"}, {512, 0, 0, "num_records 512: lanes 32.. out"},
        {1024, 512, 0, "num_records 1024, soffset 512 (lanes 32.. reach past num_records through soffset)"},
        {2048, 1024, 0, "num_records 2048, soffset 1024"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, c.nrec, c.soff, c.mode, o);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("%s:\n  ", c.what);
        for (int l : {0, 1, 2, 3, 31, 32, 33, 62, 63}) printf("lane%d=%x ", l, r[l * 4]);
        printf("\n");
    }
    return 0;
}
