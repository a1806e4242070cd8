// memguard.h -- every device / mapped-host allocation of the library goes through here.
//
// The reference runs its kernels' tests under `cuda-memcheck --leak-check full` (libzen/CMakeLists.txt:56-73, :81-86).
// This pool has no GPU memcheck (no ASAN, no XNACK), so the library carries its own two mechanisms:
//
//  * RED ZONES (any build; ZEN_HIP_REDZONE=<bytes>, a multiple of 256, in the environment before the first allocation;
//    the tests run with 4096).  Every allocation grows by that many bytes on either side, filled with a quiet-NaN
//    pattern; ZEN_HIP_POISON=1 fills the interior with another NaN pattern (a kernel that reads what nobody wrote then
//    produces NaNs instead of whatever the allocator left there).  zh_free and zen_hip_memcheck compare the zones with
//    the pattern: an out-of-bounds WRITE is reported with the allocation, the side and the offset; an out-of-bounds
//    READ shows up as a NaN in the results the parity tests compare.
//
//  * BOUNDS CHECKS (-DZEN_HIP_BOUNDS builds: ZEN_HIP_EXTRA_FLAGS=-DZEN_HIP_BOUNDS ZEN_HIP_VARIANT=bounds python
//    zen_amd/build.py).  The load / store sites of the kernels carry ZH_CHK(pointer, elements) (bounds.h), which looks
//    the address up in a device-side table of the live allocations' user ranges (kept here, red zones between them)
//    and records the first violations -- address, bytes, source line, kernel file tag -- in a host-mapped record;
//    with ZEN_HIP_BOUNDS_TRAP=1 the wavefront then executes __builtin_trap().  The shipped build compiles ZH_CHK away.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace zen_hip_impl {

hipError_t zh_malloc(void** p, size_t bytes);                      // hipMalloc
hipError_t zh_free(void* p);                                       // hipFree (also of pointers zh_malloc never saw)
hipError_t zh_host_malloc(void** p, size_t bytes, unsigned flags); // hipHostMalloc
hipError_t zh_host_device_pointer(void** dev, void* host);         // hipHostGetDevicePointer of a zh_host_malloc pointer
hipError_t zh_host_free(void* p);                                  // hipHostFree
hipError_t zh_ext_malloc(void** p, size_t bytes, unsigned flags);  // hipExtMallocWithFlags (freed with zh_free)

struct ZhRange {
	uintptr_t begin, end; // user range of a live allocation
};
struct ZhViolation {
	unsigned long long addr;
	unsigned long long tag; // the first eight characters of the source file's name
	unsigned bytes, line;
};
constexpr unsigned long long zh_file_tag(const char* path)
{
	const char* b = path;
	for (const char* c = path; *c; ++c)
		if (*c == '/')
			b = c + 1;
	unsigned long long t = 0;
	for (int i = 0; i < 8 && b[i] && b[i] != '.'; ++i)
		t |= (unsigned long long)(unsigned char)b[i] << (8 * i);
	return t;
}
struct ZhTable { // device memory; `fail` points at host-mapped memory
	unsigned n, trap;
	struct ZhFail* fail;
	ZhRange r[1]; // n entries, sorted by begin (capacity ZH_TABLE_CAP)
};
struct ZhFail {
	unsigned count;         // violations seen (atomic)
	unsigned pad[3];
	ZhViolation first[16];  // the first sixteen
};
constexpr unsigned ZH_TABLE_CAP = 16384;

// -DZEN_HIP_BOUNDS builds: every translation unit with kernels keeps its own pointer to the table (no relocatable
// device code: a __device__ variable is private to its object file) and registers a setter for it at load time.
void zh_register_table_user(void (*set)(const ZhTable*));

} // namespace zen_hip_impl
