// rt_trace_probes.h - diagnostic probes of the trace kernel, for the tuning variants only (tools/build_variants.py
// passes the -D flags).  The default build defines none of the switches below and every macro expands to nothing.
//
//   -DRT_STAMPS            shader-clock cycles per section of the main loop, summed per wavefront and added to the
//                          counter array behind RT_COUNTER_CNT (counter launches only; tools/quickbench.py prints them)
//   -DRT_X_VALU=n          n extra vector additions per node step of the walk over pairs, in 8 independent chains
//   -DRT_X_CMP=n           n / 2 extra compare + select pairs per node step
//   -DRT_X_LOAD=n          n extra 16-byte loads per node step from the cache line that has just arrived
//     -DRT_X_LOAD_UNIFORM    ... every lane reads the same line
//     -DRT_X_LOAD_ONE_LANE   ... only the first lane of the node step makes them
//   -DRT_X_SALU=n          n extra scalar additions per node step
//
// The macros use the trace kernel's local names (o, inv, tMax, L0, L1, p, blob4, lane, nodeMask, level, P, ANY, COUNT).
#ifndef RT_TRACE_PROBES_H
#define RT_TRACE_PROBES_H

#ifdef RT_STAMPS
#define RT_PROBE_STATE()                         \
    unsigned stampAcc[16] = {}, stampCnt[16] = {}; \
    unsigned long long stampLast = __builtin_amdgcn_s_memtime()
#define STAMP(i)                                                          \
    do {                                                                  \
        if (COUNT) {                                                      \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            stampAcc[i] += (unsigned)(now_ - stampLast);                  \
            stampCnt[i] += 1u;                                            \
            stampLast = now_;                                             \
        }                                                                 \
    } while (0)
#define RT_PROBE_FLUSH()                                                                                         \
    do {                                                                                                         \
        if (lane == 0)                                                                                           \
            for (int i_ = 0; i_ < 16; ++i_) {                                                                    \
                atomicAdd(P.counters + 64 + (ANY ? 32 : 0) + 2 * i_, (unsigned long long)stampAcc[i_]);          \
                atomicAdd(P.counters + 64 + (ANY ? 32 : 0) + 2 * i_ + 1, (unsigned long long)stampCnt[i_]);      \
            }                                                                                                    \
    } while (0)
#else
#define RT_PROBE_STATE() \
    do {                 \
    } while (0)
#define STAMP(i) \
    do {         \
    } while (0)
#define RT_PROBE_FLUSH() \
    do {                 \
    } while (0)
#endif

#ifdef RT_X_VALU
#define RT_PROBE_X_VALU()                                                                                                     \
    do {                                                                                                                      \
        float x0 = o.x, x1 = o.y, x2 = o.z, x3 = inv.x, x4 = inv.y, x5 = inv.z, x6 = tMax, x7 = L0.x;                          \
        _Pragma("unroll") for (int i_ = 0; i_ < RT_X_VALU / 8; ++i_)                                                          \
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"        \
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"          \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                     \
                         : "v"(L1.x));                                                                                        \
        asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));                            \
    } while (0)
#else
#define RT_PROBE_X_VALU() \
    do {                  \
    } while (0)
#endif

#ifdef RT_X_CMP
#define RT_PROBE_X_CMP()                                                                                                                    \
    do {                                                                                                                                    \
        float x0 = o.x, x1 = o.y;                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < RT_X_CMP / 2; ++i_)                                                                         \
            asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(x0), "+v"(x1) : "v"(L1.x) : "vcc");              \
        asm volatile("" ::"v"(x0), "v"(x1));                                                                                                \
    } while (0)
#else
#define RT_PROBE_X_CMP() \
    do {                 \
    } while (0)
#endif

#ifdef RT_X_LOAD
#ifdef RT_X_LOAD_UNIFORM
#define RT_PROBE_X_LOAD_FROM() (blob4 + ((size_t)(lane & 3) << 2))
#else
#define RT_PROBE_X_LOAD_FROM() (p)
#endif
#ifdef RT_X_LOAD_ONE_LANE
#define RT_PROBE_X_LOAD_IF() if (lane == __ffsll((long long)nodeMask) - 1)
#else
#define RT_PROBE_X_LOAD_IF()
#endif
#define RT_PROBE_X_LOAD()                                                \
    do {                                                                 \
        const float4* q_ = RT_PROBE_X_LOAD_FROM();                       \
        asm volatile("" : "+v"(q_));                                     \
        RT_PROBE_X_LOAD_IF()                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < RT_X_LOAD; ++i_) {       \
            const float4 x_ = q_[i_ & 3];                                \
            asm volatile("" ::"v"(x_.x), "v"(x_.y), "v"(x_.z), "v"(x_.w)); \
        }                                                                \
    } while (0)
#else
#define RT_PROBE_X_LOAD() \
    do {                  \
    } while (0)
#endif

#ifdef RT_X_SALU
#define RT_PROBE_X_SALU()                                                                                              \
    do {                                                                                                               \
        unsigned s0_ = (unsigned)level;                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < RT_X_SALU; ++i_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0_) : : "scc"); \
        asm volatile("" ::"s"(s0_));                                                                                   \
    } while (0)
#else
#define RT_PROBE_X_SALU() \
    do {                  \
    } while (0)
#endif

#define RT_PROBE_NODE_STEP() \
    do {                     \
        RT_PROBE_X_VALU();   \
        RT_PROBE_X_CMP();    \
        RT_PROBE_X_LOAD();   \
        RT_PROBE_X_SALU();   \
    } while (0)

#endif
