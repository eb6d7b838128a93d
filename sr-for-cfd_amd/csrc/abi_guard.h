// HIP-free piece of the ABI plumbing (capi_io.cpp and coarse_solver.cpp are host-only translation units, also built into the
// sanitizer harness by g++).
#pragma once
#include <exception>
#include <new>
#include <string>

#include "../../include/srcfd.h"

namespace srcfd {

void set_error(const std::string& m);

// "Never throws across the ABI" (include/srcfd.h; SURVEY.md 8b, errors): every extern "C" entry point whose body can allocate or
// call into code that throws runs inside this -- an exception becomes a status code and a message in srcfd_last_error().
template <class F>
inline int abi_guard(const char* who, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try { set_error(std::string(who) + ": out of host memory"); } catch (...) {}
    return SRCFD_ENOMEM;
  } catch (const std::exception& e) {
    try { set_error(std::string(who) + ": " + e.what()); } catch (...) {}
    return SRCFD_EINVAL;
  } catch (...) {
    try { set_error(std::string(who) + ": unknown exception"); } catch (...) {}
    return SRCFD_EINVAL;
  }
}

}  // namespace srcfd
