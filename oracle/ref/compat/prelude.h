/*
 * Dialect adapter used only when compiling the reference's own sources into
 * oracle/_ref/libref.so (test infrastructure, never shipped).
 *
 * The reference is written against MSVC: it spells the C float math functions
 * as std::sqrtf / std::powf / ... and relies on MSVC's <cmath>, <limits>,
 * <algorithm> and <string> being pulled in transitively.  glibc/libstdc++ has
 * every one of those functions and headers; this file only makes the names
 * visible where the reference expects them.  No function is re-implemented.
 */
#pragma once
#include <math.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>

namespace std {
using ::cosf;
using ::fabsf;
using ::powf;
using ::sinf;
using ::sqrtf;
} // namespace std
