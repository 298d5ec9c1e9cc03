// STAND-IN for the Teensy core's Arduino.h (tools/ref_shim/README.md): what AudioSDR.h / .cpp use of it.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#undef PI
#define PI 3.1415926535897932384626433832795
#define HALF_PI 1.5707963267948966192313216916398
#define TWO_PI 6.283185307179586476925286766559
typedef bool boolean;
typedef uint8_t byte;
#undef abs
#define abs(x) ({ __typeof__(x) _x = (x); (_x > 0) ? _x : -_x; })
struct SerialShim {
  template <typename... T> void print(T...) {}
  template <typename... T> void println(T...) {}
  template <typename... T> void begin(T...) {}
  template <typename... T> void printf(T...) {}
};
extern SerialShim Serial;
