// devvec.h -- owning device array; stands where the reference uses thrust::device_vector<T>
// (libzen/core.h:26-27) but only does what the wrappers need: allocate, zero, upload, download.
#ifndef ZG_DEVVEC_H
#define ZG_DEVVEC_H

#include <cstddef>
#include <vector>

#include <libzen/device_ptr.h>
#include <libzen/zen.h>
#include <zen_hip.h>

namespace zen {
namespace internal {
	template <typename T>
	class device_vector {
	public:
		device_vector()
		    : p_(nullptr)
		    , n_(0)
		{
		}
		explicit device_vector(std::size_t n)
		    : p_(nullptr)
		    , n_(n)
		{
			void* p = nullptr;
			throw_or_die(zen_hip_malloc(&p, n * sizeof(T)), "device_vector");
			p_ = static_cast<T*>(p);
			throw_or_die(zen_hip_memset(p_, 0, n * sizeof(T), nullptr), "device_vector");
		}
		device_vector(const std::vector<T>& host)
		    : device_vector(host.size())
		{
			assign(host);
		}
		device_vector(const device_vector&) = delete;
		device_vector& operator=(const device_vector&) = delete;
		device_vector(device_vector&& o) noexcept
		    : p_(o.p_)
		    , n_(o.n_)
		{
			o.p_ = nullptr;
			o.n_ = 0;
		}
		~device_vector() { zen_hip_free(p_); }

		std::size_t size() const { return n_; }
		thrust::device_ptr<T> data() const { return thrust::device_ptr<T>(p_); }
		T* raw() const { return p_; }

		void assign(const std::vector<T>& host)
		{
			throw_or_die(zen_hip_memcpy_h2d(p_, host.data(), (host.size() < n_ ? host.size() : n_) * sizeof(T)),
			             "device_vector::assign");
		}
		std::vector<T> to_host() const
		{
			std::vector<T> h(n_);
			throw_or_die(zen_hip_memcpy_d2h(h.data(), p_, n_ * sizeof(T)), "device_vector::to_host");
			return h;
		}

	private:
		T* p_;
		std::size_t n_;
	};
} // namespace internal
} // namespace zen

#endif
