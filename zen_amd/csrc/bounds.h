// bounds.h -- ZH_CHK(pointer, elements): the bounds check of -DZEN_HIP_BOUNDS builds (memguard.h); nothing otherwise.
#pragma once
#include "memguard.h"

#ifdef ZEN_HIP_BOUNDS
#if defined(__HIPCC__)
namespace zen_hip_impl {
namespace {
__device__ const ZhTable* zh_table_dev = nullptr;
void zh_set_table_here(const ZhTable* t) { (void)hipMemcpyToSymbol(HIP_SYMBOL(zh_table_dev), &t, sizeof(t)); }
struct ZhTableUser {
	ZhTableUser() { zh_register_table_user(&zh_set_table_here); }
};
const ZhTableUser zh_table_user_here;

__device__ __attribute__((noinline)) void zh_bounds_fail(const ZhTable* t, uintptr_t a, unsigned bytes, unsigned line, unsigned long long tag)
{
	ZhFail* f = t->fail;
	const unsigned k = __hip_atomic_fetch_add(&f->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	if (k < 16) {
		f->first[k].addr = a;
		f->first[k].bytes = bytes;
		f->first[k].line = line;
		f->first[k].tag = tag;
		__threadfence_system();
	}
	if (t->trap)
		__builtin_trap();
}

__device__ __forceinline__ void zh_bounds(const void* p, unsigned bytes, unsigned line, unsigned long long tag)
{
	const ZhTable* t = zh_table_dev;
	if (!t)
		return;
	const uintptr_t a = (uintptr_t)p;
	int lo = 0, hi = (int)t->n; // last range with begin <= a
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (t->r[mid].begin <= a)
			lo = mid;
		else
			hi = mid;
	}
	if (t->n == 0 || a < t->r[lo].begin || a + bytes > t->r[lo].end)
		zh_bounds_fail(t, a, bytes, line, tag);
}
} // namespace
} // namespace zen_hip_impl
#define ZH_CHK(p, n)                                                                                          \
	do {                                                                                                      \
		constexpr unsigned long long zh_tag__ = ::zen_hip_impl::zh_file_tag(__FILE__);                        \
		::zen_hip_impl::zh_bounds((const void*)(p), (unsigned)(sizeof(*(p)) * (n)), __LINE__, zh_tag__);      \
	} while (0)
#define ZH_CHK_BYTES(p, b)                                                                  \
	do {                                                                                    \
		constexpr unsigned long long zh_tag__ = ::zen_hip_impl::zh_file_tag(__FILE__);      \
		::zen_hip_impl::zh_bounds((const void*)(p), (unsigned)(b), __LINE__, zh_tag__);     \
	} while (0)
#endif
#else
#define ZH_CHK(p, n) ((void)0)
#define ZH_CHK_BYTES(p, b) ((void)0)
#endif
