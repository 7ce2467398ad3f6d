// Calibration of rocprofv3's FETCH_SIZE on the access pattern of the GLOBAL-mode kernels: uniformly random 8-byte gathers.
// MI355X_MICROARCH.md calibrates the counter only for wide coalesced streams (x2 on gfx950) and says "other access widths ...
// uncalibrated: calibrate on a known byte count in your own access pattern".  This kernel issues a KNOWN number of random 8-byte
// loads over a table of a chosen size (inside / far beyond the 256 MiB Infinity Cache) plus, for reference, the coalesced
// 16-B-per-lane stream of the same table.  scripts/calibrate_fetch.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and
// profiles/r02_fetch_calibration.json keeps counter / known bytes for each case.
//   usage: gather_calib <table MiB> <gathers per lane> <mode: 0 random 8 B, 1 coalesced 16 B>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather8(const long long* __restrict__ table, size_t n_elems, int per_lane, long long* out) {
  unsigned long long x = (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  long long acc = 0;
  for (int i = 0; i < per_lane; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;  // xorshift: independent uniformly random indices per lane
    acc += table[x % n_elems];
  }
  if (acc == 0x7fffffffffffffffll) out[0] = acc;  // never true: keeps the loads alive
}

__global__ void stream16(const int4* __restrict__ table, size_t n_vec, long long* out) {
  long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const int4 v = table[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x7fffffffffffffffll) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 64;
  const int per_lane = argc > 2 ? atoi(argv[2]) : 256;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const size_t bytes = mib << 20, n_elems = bytes / 8;
  long long *table = nullptr, *out = nullptr;
  if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  (void)hipMemset(table, 1, bytes);
  (void)hipDeviceSynchronize();
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  if (mode == 0) gather8<<<blocks, threads>>>(table, n_elems, per_lane, out);
  else stream16<<<blocks, threads>>>(reinterpret_cast<const int4*>(table), bytes / 16, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double useful = mode == 0 ? (double)blocks * threads * per_lane * 8.0 : (double)bytes;
  const double accesses = mode == 0 ? (double)blocks * threads * per_lane : (double)bytes / 16;
  printf("{\"mode\": \"%s\", \"table_mib\": %zu, \"accesses\": %.0f, \"useful_bytes\": %.0f, \"ms\": %.3f, \"useful_gbps\": %.1f}\n",
         mode == 0 ? "random 8 B gather" : "coalesced 16 B stream", mib, accesses, useful, ms, useful / ms / 1e6);
  return 0;
}
