// Host twin of apex_amd/csrc/gfx950/dynamic_lds.h: the dynamic LDS segment of the running workgroup (hip/hip_runtime.h g_dynsmem).  No pointer is cached: the segment
// belongs to whichever host thread runs the workgroup.  Test infrastructure only.
#pragma once
#include <hip/hip_runtime.h>
namespace hipemu { template <class T> struct LdsRef { template <class U> operator U*() const { return (U*)g_dynsmem; } T* operator+(long i) const { return (T*)g_dynsmem + i; } T& operator[](long i) const { return ((T*)g_dynsmem)[i]; } }; }
#define APX_DYNAMIC_LDS(T, name, alignment) static constexpr hipemu::LdsRef<T> name {}
