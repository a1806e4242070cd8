// win.h -- analysis windows.  Same formulas as the reference's libzen/win.h:21-51 (periodic von Hann and
// its square root, float math, PI = 3.14159265359F); evaluated on the host and uploaded once, whereas the
// reference's GPU variant writes the device vector element by element through device_reference.
#ifndef ZG_WIN_INTERNAL_H
#define ZG_WIN_INTERNAL_H

#include <cmath>
#include <cstddef>
#include <vector>

#include <devvec.h>

namespace zen {
namespace internal {
	namespace win {
		static constexpr float PI = 3.14159265359F;

		enum WindowType {
			SqrtVonHann,
			VonHann,
		};

		inline std::vector<float> make_window(WindowType type, std::size_t window_size)
		{
			std::vector<float> w(window_size, 0.0F);
			const float N = (float)window_size; // 'periodic' (matlab) form: divide by N, not N-1
			for (std::size_t n = 0; n < window_size; ++n) {
				const float hann = 0.5F * (1.0F - cosf(2.0F * PI * (float)n / N));
				w[n] = (type == SqrtVonHann) ? sqrtf(hann) : hann;
			}
			return w;
		}

		class WindowCPU {
		public:
			std::vector<float> window;
			WindowCPU(WindowType type, std::size_t window_size)
			    : window(make_window(type, window_size))
			{
			}
		};

		class WindowGPU {
		public:
			std::vector<float> host; // kept: COLA is accumulated on the host (hps.h:270-274)
			zen::internal::device_vector<float> window;
			WindowGPU(WindowType type, std::size_t window_size)
			    : host(make_window(type, window_size))
			    , window(host)
			{
			}
		};
	} // namespace win
} // namespace internal
} // namespace zen

#endif // ZG_WIN_INTERNAL_H
