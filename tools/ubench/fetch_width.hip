// FETCH_SIZE / WRITE_SIZE calibration by access width: the guide's "FETCH_SIZE x 2" correction is stated for 16-byte-per-lane streaming reads; K_ltb reads
// 8 bytes per lane.  Each kernel streams the SAME known number of bytes once (coalesced, lane-contiguous) with 4 / 8 / 16 bytes per lane and writes one word per wave, so
// under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE for the store kernels) the counter per kernel over the known bytes gives the factor for that width.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_width tools/ubench/fetch_width.hip;  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench/fetch_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template<typename T> __global__ void read_w(const T *p, size_t n, uint32_t *sink){
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t stride = (size_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for(; i < n; i += stride){ T v = p[i]; const uint32_t *w = (const uint32_t*)&v; for(unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k]; }
	if(acc == 0x12345678u) sink[0] = acc;
}
template<typename T> __global__ void write_w(T *p, size_t n){
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t stride = (size_t)gridDim.x * blockDim.x;
	T v; uint32_t *w = (uint32_t*)&v; for(unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = (uint32_t)i + k;
	for(; i < n; i += stride) p[i] = v;
}
int main(){
	const size_t bytes = (size_t)4 << 30; void *buf; uint32_t *sink;
	if(hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess){ printf("hipMalloc failed\n"); return 1; }
	hipMemset(buf, 1, bytes); hipDeviceSynchronize();
	const dim3 g(256 * 32), b(256);
	for(int rep = 0; rep < 2; rep++){
		hipLaunchKernelGGL((read_w<uint32_t>), g, b, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
		hipLaunchKernelGGL((read_w<uint2>), g, b, 0, 0, (const uint2*)buf, bytes / 8, sink);
		hipLaunchKernelGGL((read_w<uint4>), g, b, 0, 0, (const uint4*)buf, bytes / 16, sink);
		hipLaunchKernelGGL((write_w<uint32_t>), g, b, 0, 0, (uint32_t*)buf, bytes / 4);
		hipLaunchKernelGGL((write_w<uint2>), g, b, 0, 0, (uint2*)buf, bytes / 8);
		hipLaunchKernelGGL((write_w<uint4>), g, b, 0, 0, (uint4*)buf, bytes / 16);
	}
	hipDeviceSynchronize();
	printf("each kernel moved %zu bytes once\n", bytes);
	return 0;
}
