/*
 * gpu_stall.hip -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_parity.py, scripts/r6/).
 *
 * Keeps a HIP stream busy for a stated time: a one-wavefront kernel that sleeps until the device's wall clock has
 * advanced by `ms` milliseconds.  The tests hold the NULL stream busy with it while a context is created and used:
 * whatever the library still enqueues on the null stream then runs late, and anything that depended on it having run
 * shows (the context cursors were once zeroed by null-stream fills).  Loaded next to libbowtie_amd.so in
 * the same process; shares its HIP runtime, hence its null stream.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void gpu_stall_kernel(unsigned long long ticks)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

/* stream NULL = the null stream.  Returns 0, or the HIP error. */
extern "C" int gpu_stall(void* stream, unsigned ms)
{
	int dev = 0, khz = 0;
	if (hipGetDevice(&dev) != hipSuccess) return -1;
	if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;   /* 100 MHz on CDNA3/4 */
	hipLaunchKernelGGL(gpu_stall_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)ms * (unsigned long long)khz);
	return (int)hipGetLastError();
}

/* how long a call takes on the host while the null stream is busy for `ms`: hipMemset of `bytes` on the null stream
 * (what = 0), or hipMemcpy host-to-device (what = 1).  Microseconds, or -1.  (Is the call done when it returns?) */
extern "C" long long gpu_stall_probe(int what, unsigned ms, size_t bytes)
{
	void* p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) return -1;
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	if (gpu_stall(nullptr, ms) != 0) return -1;
	timespec a, b;
	clock_gettime(CLOCK_MONOTONIC, &a);
	hipError_t e;
	if (what == 0) e = hipMemset(p, 0, bytes);
	else { void* h = calloc(1, bytes); e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice); free(h); }
	clock_gettime(CLOCK_MONOTONIC, &b);
	(void)hipDeviceSynchronize();
	(void)hipFree(p);
	if (e != hipSuccess) return -1;
	return (long long)(b.tv_sec - a.tv_sec) * 1000000ll + (b.tv_nsec - a.tv_nsec) / 1000;
}
