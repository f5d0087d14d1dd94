// sph_kernels.h -- SPH smoothing kernels as device functions.
//
// Hand-written equivalents of the kernel classes in pysph/base/kernels.py
// (CubicSpline :29-163, WendlandQuintic :274-380, QuinticSpline :1050-1210,
// Gaussian :830-930).  The polynomial forms and branch conditions are the
// reference's; what changes is that h1 = 1/h, q and the normalisation
// fac = sigma * h1^dim are computed ONCE per pair and shared by W, dW/dq and
// the gradient (the reference recomputes them in every method).
// INSUP = true: the caller guarantees q < radius_scale (uniform-h pairs that
// passed the neighbour criterion), so the support test is dropped.
#pragma once

#include <hip/hip_runtime.h>

template <int KIND> struct SphKernel;

// max / min as ONE instruction: fmax()/fmin() compile to a canonicalising v_max x, x in front of the v_max proper
// (IEEE mode, signalling-NaN quieting); the operands here are results of arithmetic, canonical already
__device__ __forceinline__ double raw_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double raw_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float raw_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float raw_min(float a, float b) { return fminf(a, b); }


// kernels.py:73-79: fac = self.fac*h1 | *h1*h1 | *h1*h1*h1
template <class R> __device__ __forceinline__ R kernel_norm(R sigma, R h1, int dim)
{
    R f = sigma * h1;
    if (dim > 1) f *= h1;
    if (dim > 2) f *= h1;
    return f;
}

template <> struct SphKernel<1> { // CubicSpline
    static constexpr bool HAS_DWQ = false;
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dwq(R) { return R(0); }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R w(R q)
    {
        R t2 = R(2) - q;
        R a = R(0.25) * t2 * t2 * t2;
        R b = R(1) - R(1.5) * q * q * (R(1) - R(0.5) * q);
        return (!INSUP && q > R(2)) ? R(0) : (q > R(1) ? a : b);
    }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dw(R q)
    {
        R t2 = R(2) - q;
        R a = R(-0.75) * t2 * t2;
        R b = R(-3) * q * (R(1) - R(0.75) * q);
        return (!INSUP && q > R(2)) ? R(0) : (q > R(1) ? a : b);
    }
};

template <> struct SphKernel<2> { // WendlandQuintic
    static constexpr bool HAS_DWQ = true; // dw/q = -5 (1 - q/2)^3: no division by r needed
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dwq(R q)
    {
        R t = R(1) - R(0.5) * q;
        return (INSUP || q < R(2)) ? R(-5) * t * t * t : R(0);
    }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R w(R q)
    {
        R t = R(1) - R(0.5) * q;
        R v = t * t * t * t * (R(2) * q + R(1));
        return (INSUP || q < R(2)) ? v : R(0);
    }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dw(R q)
    {
        R t = R(1) - R(0.5) * q;
        R v = R(-5) * q * t * t * t;
        return (INSUP || q < R(2)) ? v : R(0);
    }
};

template <> struct SphKernel<3> { // QuinticSpline
    static constexpr bool HAS_DWQ = false;
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dwq(R) { return R(0); }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R w(R q)
    {
        // the branches q <= 2, q <= 1 of kernels.py:1120-1140 as max(., 0): a term beyond its knot is exactly zero,
        // so the value is the reference's bit for bit, without two compare-and-select pairs
        R t3 = R(3) - q, t2 = raw_max(R(2) - q, R(0)), t1 = raw_max(R(1) - q, R(0));
        R v = t3 * t3 * t3 * t3 * t3;
        v -= R(6) * t2 * t2 * t2 * t2 * t2;
        v += R(15) * t1 * t1 * t1 * t1 * t1;
        return (!INSUP && q > R(3)) ? R(0) : v;
    }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dw(R q)
    {
        R t3 = R(3) - q, t2 = raw_max(R(2) - q, R(0)), t1 = raw_max(R(1) - q, R(0));
        R v = R(-5) * t3 * t3 * t3 * t3;
        v += R(30) * t2 * t2 * t2 * t2;
        v -= R(75) * t1 * t1 * t1 * t1;
        return (!INSUP && q > R(3)) ? R(0) : v;
    }
};

// The Gaussian is CUT OFF at q = 3 where it is still 1.2e-4 of its peak, so the
// support test is kept even when the caller guarantees the neighbour criterion:
// a pair at r = 3h to within rounding must give exactly 0 like the reference
// (the polynomial kernels vanish at their support radius, there INSUP is safe).
template <> struct SphKernel<4> { // Gaussian
    static constexpr bool HAS_DWQ = true; // dw/q = -2 exp(-q^2)
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dwq(R q) { return (q < R(3)) ? R(-2) * exp(-q * q) : R(0); }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R w(R q) { return (q < R(3)) ? exp(-q * q) : R(0); }
    template <bool INSUP = false, class R> static __device__ __forceinline__ R dw(R q)
    {
        return (q < R(3)) ? R(-2) * q * exp(-q * q) : R(0);
    }
};
