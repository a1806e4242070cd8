// libzen/zen.h -- library-wide basics.  Interface-compatible with the reference's
// libzen/libzen/zen.h:6-18 (zen::ZgException, zen::Backend, zen::Eps).
#ifndef ZG_PUB_H
#define ZG_PUB_H

#include <limits>
#include <stdexcept>
#include <string>

namespace zen {

// which implementation a policy template selects (libzen/core.h TypeTraits)
enum Backend { GPU, CPU };

// thrown for caller mistakes: filter longer than the matrix, hop_h not a multiple of hop_p
class ZgException : public std::runtime_error {
public:
	explicit ZgException(const std::string& what_arg)
	    : std::runtime_error(what_arg)
	{
	}
};

constexpr float Eps = std::numeric_limits<float>::epsilon();

// C-ABI status -> the reference's error conventions (SURVEY 8(b) "Error conventions"):
// programmer errors throw ZgException, resource/runtime errors print and exit.
void throw_or_die(int zen_hip_status, const char* where);

} // namespace zen

#endif /* ZG_PUB_H */
