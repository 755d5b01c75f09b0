/*
 * rt_march.h -- the device functions every form of the trace kernel is made
 * of: vector loads / stores of R rays per lane, the rows of one element, and
 * the march over the elements with the state in registers.
 */
#ifndef RT_MARCH_H
#define RT_MARCH_H

#include <hip/hip_runtime.h>
#include "rt_math.h"
#include "rt_lay.h"

#define RT_BLOCK 256 /* rays per workgroup of the trace kernels: 4 wavefronts */

template <int R> struct rt_vec;
template <> struct rt_vec<1> { typedef double type; };
template <> struct rt_vec<2> {
    typedef double type __attribute__((ext_vector_type(2)));
};
template <> struct rt_vec<4> {
    typedef double type __attribute__((ext_vector_type(4)));
};

template <int R>
__device__ __forceinline__ void rt_load(const double *__restrict__ p,
                                        double (&v)[R])
{
    typedef typename rt_vec<R>::type V;
    const V x = *reinterpret_cast<const V *>(p);
    if constexpr (R == 1) {
        v[0] = x;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[r] = x[r];
    }
}

template <int R, int NT>
__device__ __forceinline__ void rt_store(double *__restrict__ p,
                                         const double (&v)[R])
{
    typedef typename rt_vec<R>::type V;
    V x;
    if constexpr (R == 1) {
        x = v[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            x[r] = v[r];
    }
    /* NT: 0 ordinary store, 1 non-temporal; 2 / 3 / 4 (measurements, one
     * ray per lane only): "sc1 nt", "sc0 sc1 nt", "sc1" */
    if constexpr (NT == 1)
        __builtin_nontemporal_store(x, reinterpret_cast<V *>(p));
    else if constexpr (NT == 2 && R == 1)
        asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(x)
                     : "memory");
    else if constexpr (NT == 3 && R == 1)
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p),
                     "v"(x)
                     : "memory");
    else if constexpr (NT == 4 && R == 1)
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(x)
                     : "memory");
    else
        *reinterpret_cast<V *>(p) = x;
}

/* read rows start-1 of Y,U for the R rays at column `col` */
template <int R>
__device__ __forceinline__ void rt_load_state(const rt_lay &a, int srow,
                                              int64_t col, double (&y)[R][3],
                                              double (&u)[R][3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p[R], q[R];
        rt_load<R>(a.Y + srow * a.ss + c * a.cs + col, p);
        rt_load<R>(a.U + srow * a.ss + c * a.cs + col, q);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            y[r][c] = p[r];
            u[r][c] = q[r];
        }
    }
}

/* the rows of one element for the R rays at column `col` */
template <int R, int NT>
__device__ __forceinline__ void rt_store_rows(
    unsigned flags, int s, const rt_lay &a, int64_t col,
    const double (&y)[R][3], const double (&u)[R][3],
    const double (&iv)[R][3], const double (&t)[R])
{
    if (flags & RT_F_NOSTORE)
        return;
    const int64_t row = s * a.ss + col;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p[R], q[R], d[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            p[r] = y[r][c];
            q[r] = u[r][c];
            d[r] = iv[r][c];
        }
        rt_store<R, NT>(a.Y + row + c * a.cs, p);
        if (!(flags & RT_F_SKIP_U))
            rt_store<R, NT>(a.U + row + c * a.cs, q);
        if (flags & RT_F_STORE_I)
            rt_store<R, NT>(a.I + row + c * a.cs, d);
    }
    rt_store<R, NT>(a.T + s * a.ssT + col, t);
}

/* all elements start..stop-1 for the R rays of this lane; state in VGPRs */
/*
 * ASPH = false: the table holds no aspheric element (the host knows: it
 * finalises the flags).  Their bits are masked out of `flags` HERE, where the
 * compiler can see it, so that everything behind them -- both Newton solves,
 * the fast refraction, the aspheric normals -- is not in that instantiation
 * at all: the kernel of a system of planes, spheres and conics is a third of
 * the code, and what the compiler makes of it no longer moves when the
 * asphere code is touched (round 6: 153 -> 238 VALU instructions per
 * ray-surface op in the generated-batch kernel of the double-Gauss from a
 * change inside the Newton loop it never runs).
 */
#define RT_FLAGS_OF(S, ASPH)                                                  \
    ((ASPH) ? (S)->flags : ((S)->flags & ~(RT_F_ASPH | RT_F_FAST)))

template <int R, int NT, bool ASPH = true>
__device__ __forceinline__ void rt_march(const rt_surface *__restrict__ surf,
                                         int start, int stop, int clip,
                                         const rt_lay &a, int64_t col,
                                         double (&y)[R][3],
                                         double (&u)[R][3])
{
    double iv[R][3], t[R];
    {
        const rt_surface *S0 = surf + (start - 1);
        rt_leave<R>(S0, S0->flags, y, u);
    }
    for (int s = start; s < stop; ++s) {
        const rt_surface *S = surf + s;
        const unsigned flags = RT_FLAGS_OF(S, ASPH);
        /* a ray whose direction is NaN (clipped, missed, TIR, Newton failure:
         * elements.py:206-209,:496,:367,:347) yields NaN in every array of
         * every later element; a wavefront with no other ray left stores
         * that without evaluating it */
        bool alive = false;
#pragma unroll
        for (int r = 0; r < R; ++r)
            alive = alive || u[r][0] == u[r][0];
        if (RT_WAVE_ANY(alive)) {
            rt_step<R>(S, flags, clip, y, u, iv, t);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[r] = RT_NAN;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    y[r][c] = u[r][c] = iv[r][c] = RT_NAN;
            }
        }

        /* all rows of the element leave in one burst: measured 3 % faster
         * than sending y,t,i ahead of the refraction, and aligning the waves
         * of a workgroup with a barrier first does not help
         * (profiles/r01_probes/ab_store_order.log) */
        rt_store_rows<R, NT>(flags, s, a, col, y, u, iv, t);

        rt_leave<R>(S, flags, y, u);
    }
}

#endif /* RT_MARCH_H */
