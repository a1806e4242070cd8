// libzen/device_ptr.h -- the sliver of thrust the reference's PUBLIC headers expose
// (thrust::device_ptr<float> in libzen/libzen/hps.h:96-101 and libzen/libzen/io.h:20-21), so that
// callers written against Zen compile unchanged with a plain host compiler.  If real (roc)Thrust has
// already been included these definitions step aside.
#ifndef ZG_DEVICE_PTR_H
#define ZG_DEVICE_PTR_H

#include <cstddef>

#if !defined(THRUST_VERSION)
namespace thrust {
template <typename T>
class device_ptr {
public:
	device_ptr()
	    : p_(nullptr)
	{
	}
	explicit device_ptr(T* p)
	    : p_(p)
	{
	}
	T* get() const { return p_; }
	device_ptr operator+(std::ptrdiff_t n) const { return device_ptr(p_ + n); }
	device_ptr& operator+=(std::ptrdiff_t n)
	{
		p_ += n;
		return *this;
	}
	bool operator==(const device_ptr& o) const { return p_ == o.p_; }
	bool operator!=(const device_ptr& o) const { return p_ != o.p_; }

private:
	T* p_;
};

template <typename T>
inline device_ptr<T> device_pointer_cast(T* p)
{
	return device_ptr<T>(p);
}
template <typename T>
inline T* raw_pointer_cast(const device_ptr<T>& p)
{
	return p.get();
}
} // namespace thrust
#endif

#endif /* ZG_DEVICE_PTR_H */
