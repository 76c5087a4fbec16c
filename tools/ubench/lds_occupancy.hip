// How many 64-thread workgroups of a given dynamic-LDS size run concurrently on one CU?  Each block spins ~200 us;
// grid = 256 CUs x 64 blocks; the elapsed time / spin time = number of rounds -> blocks per CU per round.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) spin(long long ticks, int *sink){
	extern __shared__ int lds[];
	lds[threadIdx.x] = threadIdx.x;
	long long t0 = clock64();
	while(clock64() - t0 < ticks) { }
	if(lds[threadIdx.x] == -1) *sink = 1;
}
int main(){
	int *sink; hipMalloc(&sink, 4);
	hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
	printf("CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, clockRate %d kHz\n", pr.multiProcessorCount, pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.clockRate);
	const int kbs[] = {0, 1, 2, 4, 8, 12, 16, 24, 32, 64};
	for(int kb : kbs){
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		const long long ticks = 20000000 / 100;   // clock64 ticks
		const int nblk = pr.multiProcessorCount * 64;
		hipLaunchKernelGGL(spin, dim3(nblk), dim3(64), kb * 1024, 0, ticks, sink);
		hipDeviceSynchronize();
		hipEventRecord(a);
		hipLaunchKernelGGL(spin, dim3(nblk), dim3(64), kb * 1024, 0, ticks, sink);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), kb * 1024, 0, ticks, sink);
		hipEventRecord(a);
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), kb * 1024, 0, ticks, sink);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms1; hipEventElapsedTime(&ms1, a, b);
		printf("LDS %2d KB: one block %.3f ms, %d blocks %.3f ms -> %.1f rounds -> %.1f blocks/CU concurrently\n", kb, ms1, nblk, ms, ms / ms1, 64.0 / (ms / ms1));
	}
	return 0;
}
