// Shared plumbing of the header-only C++ classes that mirror the reference's public interface
// (reference include/{harvest,cheaptrick,d4c,synthesis}.hpp) on top of the C-ABI in world_class_c.h.
// The reference's methods return void and never fail; here a backend failure (no HIP device, bad
// argument, ...) throws std::runtime_error carrying wc_last_error().
#ifndef WORLD_CLASS_COMMON_HPP
#define WORLD_CLASS_COMMON_HPP

#include <stdexcept>
#include <string>

#include "world_class_c.h"

namespace world_class {
namespace detail {
inline void check(int rc, const char *what) {
	if (rc != WC_OK) throw std::runtime_error(std::string(what) + ": " + wc_last_error());
}
template <class T>
inline T *checked(T *handle, const char *what) {
	if (!handle) throw std::runtime_error(std::string(what) + ": " + wc_last_error());
	return handle;
}
}  // namespace detail
}  // namespace world_class

#endif
