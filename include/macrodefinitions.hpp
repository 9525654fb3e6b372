// Drop-in for the reference's include/macrodefinitions.hpp (reference :1-144): the linkage / visibility macros a caller of the
// reference may have picked up from it.  Same names, same meaning:
//   WORLD_BEGIN_C_DECLS / WORLD_END_C_DECLS   extern "C" { ... } when compiled as C++, nothing in C
//   WORLD_API / WORLD_LOCAL                   exported / hidden symbol when WORLD_LIBRARIES_EXPORTS is defined (WORLD_SRC: building
//                                             the library rather than using it), nothing for a static build
// This library's own entry points are declared in world_class_c.h with plain extern "C".
#ifndef WORLD_MACRODEFINITIONS_HPP
#define WORLD_MACRODEFINITIONS_HPP

#undef WORLD_BEGIN_C_DECLS
#undef WORLD_END_C_DECLS
#ifdef __cplusplus
#define WORLD_BEGIN_C_DECLS extern "C" {
#define WORLD_END_C_DECLS }
#else
#define WORLD_BEGIN_C_DECLS
#define WORLD_END_C_DECLS
#endif

#if defined _WIN32 || defined __CYGWIN__
#define WORLD_HELPER_DLL_IMPORT __declspec(dllimport)
#define WORLD_HELPER_DLL_EXPORT __declspec(dllexport)
#define WORLD_HELPER_DLL_LOCAL
#elif defined __GNUC__ && __GNUC__ >= 4
#define WORLD_HELPER_DLL_IMPORT __attribute__((visibility("default")))
#define WORLD_HELPER_DLL_EXPORT __attribute__((visibility("default")))
#define WORLD_HELPER_DLL_LOCAL __attribute__((visibility("hidden")))
#else
#define WORLD_HELPER_DLL_IMPORT
#define WORLD_HELPER_DLL_EXPORT
#define WORLD_HELPER_DLL_LOCAL
#endif

#ifdef WORLD_LIBRARIES_EXPORTS
#ifdef WORLD_SRC
#define WORLD_API WORLD_HELPER_DLL_EXPORT
#else
#define WORLD_API WORLD_HELPER_DLL_IMPORT
#endif
#define WORLD_LOCAL WORLD_HELPER_DLL_LOCAL
#else
#define WORLD_API
#define WORLD_LOCAL
#endif

#endif  // WORLD_MACRODEFINITIONS_HPP
