// memguard.hip -- red zones around every allocation of the library, and the live-range table of the bounds-checking
// build (memguard.h).  Host code, and one test kernel (zen_hip_debug_poke).
#include "common.h"
#include "bounds.h"
#include "memguard.h"

#include <algorithm>
#include <cstdlib>
#include <unistd.h>
#include <map>
#include <mutex>
#include <vector>

namespace zen_hip_impl {
namespace {

constexpr unsigned GUARD_WORD = 0x7FC5A5A5u;  // a quiet NaN
constexpr unsigned POISON_WORD = 0x7FC0BEEFu; // another one

struct Rec {
	void* base;   // what the runtime returned
	size_t bytes; // user bytes
	size_t zone;  // bytes on either side
	int kind;     // 0 device, 1 mapped host
	void* dev;    // the address kernels use for the user range (mapped host memory: its device alias)
};

struct Guard {
	std::mutex mu;
	bool init = false;
	size_t zone = 0;
	bool poison = false;
	std::map<void*, Rec> live; // by user pointer
	unsigned long long corrupt_words = 0, corrupt_allocs = 0, n_allocs = 0;
	char first_msg[256] = "";
	// bounds-checking builds
	std::vector<void (*)(const ZhTable*)> users;
	ZhTable* table = nullptr; // device
	ZhFail* fail = nullptr;   // host-mapped
	ZhFail* fail_dev = nullptr;
	bool users_set = false;
	bool table_dirty = true;
	bool table_overflow_reported = false;
};
Guard& G()
{
	static Guard* g = new Guard; // (never destroyed: translation units register during their static initialisation)
	return *g;
}

#ifdef ZEN_HIP_BOUNDS
constexpr bool BOUNDS_BUILD = true;
#else
constexpr bool BOUNDS_BUILD = false;
#endif

// A process that ran with red zones and found one overwritten (or, bounds build, recorded an out-of-bounds access) does
// not exit with status 0, whatever its own verdict was: tests that start child processes see it.
void report_at_exit()
{
	Guard& g = G();
	const unsigned long long viol = g.fail ? __atomic_load_n(&g.fail->count, __ATOMIC_ACQUIRE) : 0;
	if (g.corrupt_allocs || viol) {
		fprintf(stderr, "zen_hip memguard: %llu red-zone words overwritten in %llu zones, %llu out-of-bounds accesses recorded; exit status 86\n",
		        g.corrupt_words, g.corrupt_allocs, viol);
		fflush(nullptr); // this handler was registered late and runs early: the program's buffered stdout (the CLI's report, a test's
		                 // verdict) must not be lost with the exit status that asks somebody to read it
		_exit(86);
	}
}

void lazy_init(Guard& g)
{
	if (g.init)
		return;
	g.init = true;
	const char* z = getenv("ZEN_HIP_REDZONE");
	size_t v = z ? (size_t)strtoull(z, nullptr, 10) : 0;
	if (BOUNDS_BUILD && v == 0)
		v = 4096; // the table tells allocations apart by the zones between them
	g.zone = (v + 255) / 256 * 256;
	const char* p = getenv("ZEN_HIP_POISON");
	g.poison = p && *p && *p != '0';
	if (g.zone)
		atexit(report_at_exit);
}

hipError_t fill_words(void* dst, unsigned word, size_t bytes, int kind)
{
	if (bytes == 0)
		return hipSuccess;
	if (kind == 1) {
		unsigned* w = (unsigned*)dst;
		for (size_t i = 0; i < bytes / 4; ++i)
			w[i] = word;
		return hipSuccess;
	}
	return hipMemsetD32((hipDeviceptr_t)dst, (int)word, bytes / 4);
}

void note_corruption(Guard& g, const Rec& r, const char* side, size_t first_off, size_t words)
{
	g.corrupt_words += words;
	++g.corrupt_allocs;
	char msg[256];
	snprintf(msg, sizeof(msg), "zen_hip memguard: %s allocation of %zu bytes at %p: %zu words of the %s red zone overwritten, first at byte offset %zu of the zone",
	         r.kind ? "mapped host" : "device", r.bytes, (void*)((char*)r.base + r.zone), words, side, first_off);
	fprintf(stderr, "%s\n", msg);
	if (!g.first_msg[0])
		memcpy(g.first_msg, msg, sizeof(msg));
}

// compares both zones of an allocation with the pattern (the device must be idle: the callers synchronise)
void verify(Guard& g, const Rec& r)
{
	if (r.zone == 0)
		return;
	std::vector<unsigned> buf(r.zone / 4);
	for (int side = 0; side < 2; ++side) {
		const char* z = (const char*)r.base + (side ? r.zone + ((r.bytes + 255) / 256 * 256) : 0);
		if (r.kind == 1)
			memcpy(buf.data(), z, r.zone);
		else if (hipMemcpy(buf.data(), z, r.zone, hipMemcpyDeviceToHost) != hipSuccess) {
			(void)hipGetLastError();
			continue;
		}
		size_t bad = 0, first = 0;
		for (size_t i = 0; i < buf.size(); ++i)
			if (buf[i] != GUARD_WORD) {
				if (!bad)
					first = i * 4;
				++bad;
			}
		if (bad) {
			note_corruption(g, r, side ? "back" : "front", first, bad);
			(void)fill_words(const_cast<char*>(z), GUARD_WORD, r.zone, r.kind); // (reported once)
		}
	}
	// the slack between the user's last byte and the back zone (allocations are padded to 256 bytes) is not checked:
	// the kernels' 16-byte accesses may legitimately touch the rest of a row's last vector
}

hipError_t push_table(Guard& g)
{
	if (!BOUNDS_BUILD)
		return hipSuccess;
	if (!g.table) {
		hipError_t e = hipMalloc((void**)&g.table, sizeof(ZhTable) + sizeof(ZhRange) * ZH_TABLE_CAP);
		if (e != hipSuccess)
			return e;
		e = hipHostMalloc((void**)&g.fail, sizeof(ZhFail), hipHostMallocMapped | hipHostMallocPortable);
		if (e != hipSuccess)
			return e;
		memset(g.fail, 0, sizeof(ZhFail));
		void* d = nullptr;
		e = hipHostGetDevicePointer(&d, g.fail, 0);
		if (e != hipSuccess)
			return e;
		g.fail_dev = (ZhFail*)d;
	}
	std::vector<char> host(sizeof(ZhTable) + sizeof(ZhRange) * g.live.size());
	ZhTable* t = (ZhTable*)host.data();
	const char* trap = getenv("ZEN_HIP_BOUNDS_TRAP");
	t->trap = (trap && *trap && *trap != '0') ? 1u : 0u;
	t->fail = g.fail_dev;
	unsigned n = 0;
	for (const auto& kv : g.live) {
		if (n >= ZH_TABLE_CAP) { // more live allocations than the table holds: accesses to the rest would read as violations
			if (!g.table_overflow_reported) {
				fprintf(stderr, "zen_hip memguard: %zu live allocations, the bounds table holds %u: accesses to the others will be "
				                "reported as out of bounds\n", g.live.size(), (unsigned)ZH_TABLE_CAP);
				g.table_overflow_reported = true;
			}
			break;
		}
		t->r[n].begin = (uintptr_t)kv.second.dev;
		t->r[n].end = (uintptr_t)kv.second.dev + ((kv.second.bytes + 255) / 256 * 256); // (incl. the alignment slack, see verify)
		++n;
	}
	t->n = n;
	std::sort(t->r, t->r + n, [](const ZhRange& a, const ZhRange& b) { return a.begin < b.begin; });
	(void)hipDeviceSynchronize(); // (no kernel looks at a half-written table)
	hipError_t e = hipMemcpy(g.table, host.data(), host.size(), hipMemcpyHostToDevice);
	if (e != hipSuccess)
		return e;
	if (!g.users_set) {
		for (auto set : g.users)
			set(g.table);
		g.users_set = true;
	}
	g.table_dirty = false;
	return hipSuccess;
}

hipError_t alloc_common(void** p, size_t bytes, int kind, unsigned flags, int ext)
{
	Guard& g = G();
	std::lock_guard<std::mutex> lk(g.mu);
	lazy_init(g);
	if (bytes == 0)
		bytes = 1;
	const size_t body = (bytes + 255) / 256 * 256;
	const size_t total = g.zone ? body + 2 * g.zone : bytes;
	void* base = nullptr;
	hipError_t e;
	if (kind == 1)
		e = hipHostMalloc(&base, total, flags);
	else if (ext)
		e = hipExtMallocWithFlags(&base, total, flags);
	else
		e = hipMalloc(&base, total);
	if (e != hipSuccess)
		return e;
	++g.n_allocs;
	if (g.zone == 0 && !BOUNDS_BUILD) {
		*p = base;
		return hipSuccess;
	}
	char* user = (char*)base + g.zone;
	if (g.zone) {
		e = fill_words(base, GUARD_WORD, g.zone, kind);
		if (e == hipSuccess)
			e = fill_words(user + body, GUARD_WORD, g.zone, kind);
		if (e == hipSuccess && g.poison)
			e = fill_words(user, POISON_WORD, body, kind);
		if (e != hipSuccess) {
			(void)(kind == 1 ? hipHostFree(base) : hipFree(base));
			return e;
		}
	}
	void* dev_user = user;
	if (kind == 1) {
		void* d = nullptr;
		if (hipHostGetDevicePointer(&d, base, 0) == hipSuccess)
			dev_user = (char*)d + g.zone;
		else
			(void)hipGetLastError();
	}
	g.live[user] = Rec{base, bytes, g.zone, kind, dev_user};
	*p = user;
	return push_table(g);
}

hipError_t free_common(void* p, int kind)
{
	if (!p)
		return hipSuccess;
	Guard& g = G();
	std::unique_lock<std::mutex> lk(g.mu);
	auto it = g.live.find(p);
	if (it == g.live.end()) {
		lk.unlock();
		return kind == 1 ? hipHostFree(p) : hipFree(p);
	}
	const Rec r = it->second;
	(void)hipDeviceSynchronize(); // whatever still writes to it (hipFree would wait too)
	verify(g, r);
	g.live.erase(it);
	hipError_t e = push_table(g);
	const hipError_t e2 = r.kind == 1 ? hipHostFree(r.base) : hipFree(r.base);
	return e2 != hipSuccess ? e2 : e;
}

__global__ void poke_kernel(unsigned* p, unsigned value)
{
	ZH_CHK(p, 1);
	*p = value;
}

} // namespace

hipError_t zh_malloc(void** p, size_t bytes) { return alloc_common(p, bytes, 0, 0, 0); }
hipError_t zh_ext_malloc(void** p, size_t bytes, unsigned flags) { return alloc_common(p, bytes, 0, flags, 1); }
hipError_t zh_free(void* p) { return free_common(p, 0); }
hipError_t zh_host_malloc(void** p, size_t bytes, unsigned flags) { return alloc_common(p, bytes, 1, flags, 0); }
hipError_t zh_host_free(void* p) { return free_common(p, 1); }
hipError_t zh_host_device_pointer(void** dev, void* host)
{
	Guard& g = G();
	{
		std::lock_guard<std::mutex> lk(g.mu);
		auto it = g.live.find(host);
		if (it != g.live.end() && it->second.kind == 1) {
			void* d = nullptr;
			const hipError_t e = hipHostGetDevicePointer(&d, it->second.base, 0);
			if (e != hipSuccess)
				return e;
			*dev = (char*)d + it->second.zone;
			return hipSuccess;
		}
	}
	return hipHostGetDevicePointer(dev, host, 0);
}

void zh_register_table_user(void (*set)(const ZhTable*))
{
	Guard& g = G();
	std::lock_guard<std::mutex> lk(g.mu);
	g.users.push_back(set);
}

} // namespace zen_hip_impl

using namespace zen_hip_impl;

extern "C" {

int zen_hip_memcheck(zen_hip_memcheck_report* out)
{
	if (!out)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "memcheck: null argument");
	Guard& g = G();
	std::lock_guard<std::mutex> lk(g.mu);
	lazy_init(g);
	memset(out, 0, sizeof(*out));
	out->redzone_bytes = g.zone;
	out->bounds_build = BOUNDS_BUILD ? 1 : 0;
	if (g.zone) {
		ZH_HIP(hipDeviceSynchronize());
		for (const auto& kv : g.live)
			verify(g, kv.second);
	}
	out->live_allocations = g.live.size();
	out->allocations = g.n_allocs;
	out->corrupt_words = g.corrupt_words;
	out->corrupt_allocations = g.corrupt_allocs;
	memcpy(out->first_message, g.first_msg, sizeof(out->first_message));
	if (g.fail) {
		out->bounds_violations = __atomic_load_n(&g.fail->count, __ATOMIC_ACQUIRE);
		if (out->bounds_violations) {
			const ZhViolation& v = g.fail->first[0];
			char name[9] = {0};
			for (int i = 0; i < 8; ++i)
				name[i] = (char)((v.tag >> (8 * i)) & 0xff);
			snprintf(out->first_violation, sizeof(out->first_violation), "%u bytes at %#llx from %s*:%u (of %llu out-of-bounds accesses)", v.bytes,
			         v.addr, name, v.line, (unsigned long long)out->bounds_violations);
			if (!out->first_message[0])
				snprintf(out->first_message, sizeof(out->first_message), "zen_hip memguard: out of bounds: %s", out->first_violation);
		}
	}
	return ZEN_HIP_OK;
}

int zen_hip_debug_poke(void* dev, long long byte_offset, unsigned value)
{
	if (!dev)
		ZH_FAIL(ZEN_HIP_E_BAD_ARG, "debug_poke: null argument");
	// a store to any device address: only what the memory-checking tests are for -- red zones on, or the bounds build
	{
		Guard& g = G();
		std::lock_guard<std::mutex> lk(g.mu);
		lazy_init(g);
	}
	if (!BOUNDS_BUILD && G().zone == 0)
		ZH_FAIL(ZEN_HIP_E_UNSUPPORTED, "debug_poke: a diagnostic of runs with ZEN_HIP_REDZONE (or of the -DZEN_HIP_BOUNDS build)");
	hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(1), 0, nullptr, (unsigned*)((char*)dev + byte_offset), value);
	ZH_HIP(hipGetLastError());
	ZH_HIP(hipDeviceSynchronize());
	return ZEN_HIP_OK;
}

} // extern "C"
