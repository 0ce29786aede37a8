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
//   -DRT_TOP_STATS         the walk over quads tallies, per node step, how many of its lanes sit in the top 1 / 2 / 3 / 4
//                          quad levels of their tree (the root's quad is level 0) and whether ALL of them do - what a copy
//                          of the tree's top in LDS could serve.  The level travels in bits 24-28 of the interior
//                          references (trees of fewer than 2^24 quads only); the frame's last kernel prints the sums
//
// The macros use the trace kernel's local names (o, inv, tMax, L0, L1, p, blob4, lane, nodeMask, level, P, ANY, COUNT).
#ifndef RT_TRACE_PROBES_H
#define RT_TRACE_PROBES_H

#ifdef RT_TOP_STATS
#define RT_PROBE_TOP_STATE() unsigned topSteps = 0, topLanes = 0, topIn[4] = {}, uniSteps[5] = {}, sameQuad[4] = {}
#else
#define RT_PROBE_TOP_STATE() \
    do {                     \
    } while (0)
#endif
#ifdef RT_STAMPS
#define RT_PROBE_STATE()                         \
    RT_PROBE_TOP_STATE();                          \
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
#define RT_PROBE_STATE() RT_PROBE_TOP_STATE()
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

#ifdef RT_TOP_STATS
#define RT_PROBE_QUAD_UNITS(r) ((r)&0xffffffu)
#define RT_PROBE_QUAD_VOTE()                                                                 \
    do {                                                                                     \
        if (kQuad) {                                                                         \
            const bool node_ = (int)cur >= 0;                                                \
            const unsigned lvl_ = (cur >> 24) & 31u;                                         \
            const unsigned long long act_ = __ballot(node_);                                 \
            uniSteps[4]++;                                                                   \
            {                                                                                \
                /* distinct quads among the step's lanes: 1, 2, 3-4, more */                  \
                unsigned long long left_ = act_;                                             \
                unsigned distinct_ = 0;                                                      \
                while (left_ != 0ull && distinct_ < 5u) {                                    \
                    const unsigned q_ = __builtin_amdgcn_readlane(cur & 0xffffffu, __ffsll((long long)left_) - 1); \
                    left_ &= ~__ballot(node_ && (cur & 0xffffffu) == q_);                    \
                    ++distinct_;                                                             \
                }                                                                            \
                sameQuad[distinct_ <= 1u ? 0 : (distinct_ == 2u ? 1 : (distinct_ <= 4u ? 2 : 3))]++; \
            }                                                                                \
            _Pragma("unroll") for (unsigned k_ = 0; k_ < 4; ++k_)                           \
                if (__ballot(node_ && lvl_ <= k_) == act_) uniSteps[k_]++;                   \
        }                                                                                    \
    } while (0)
#define RT_PROBE_QUAD_STEP()                                                                               \
    do {                                                                                                   \
        const unsigned lvl_ = (cur >> 24) & 31u;                                                           \
        const unsigned nxt_ = (lvl_ < 31u ? lvl_ + 1u : 31u) << 24;                                        \
        r0 |= RT_REF_IS_LEAF(r0) ? 0u : nxt_;                                                              \
        r1 |= RT_REF_IS_LEAF(r1) ? 0u : nxt_;                                                              \
        r2 |= RT_REF_IS_LEAF(r2) ? 0u : nxt_;                                                              \
        r3 |= RT_REF_IS_LEAF(r3) ? 0u : nxt_;                                                              \
        topSteps++;                                                                                        \
        topLanes += 1u; /* per lane: summed over the wavefront at the end */                               \
        _Pragma("unroll") for (unsigned k_ = 0; k_ < 4; ++k_) {                                           \
            const bool in_ = lvl_ <= k_;                                                                   \
            topIn[k_] += in_ ? 1u : 0u;                                                                    \
        }                                                                                                  \
    } while (0)
#define RT_PROBE_QUAD_FLUSH()                                                                              \
    do {                                                                                                   \
        if (kQuad) {                                                                                       \
            unsigned v_[5] = {topLanes, topIn[0], topIn[1], topIn[2], topIn[3]};                            \
            for (int i_ = 0; i_ < 5; ++i_) {                                                               \
                for (int off_ = 32; off_ > 0; off_ >>= 1) v_[i_] += __shfl_down(v_[i_], off_, kWave);      \
                if (lane == 0) atomicAdd(P.ctl + 100 + (ANY ? 16 : 0) + i_, v_[i_]);                       \
            }                                                                                              \
            if (lane == 0) {                                                                               \
                atomicAdd(P.ctl + 105 + (ANY ? 16 : 0), uniSteps[4]);                                      \
                for (int i_ = 0; i_ < 4; ++i_) atomicAdd(P.ctl + 106 + (ANY ? 16 : 0) + i_, uniSteps[i_]); \
                for (int i_ = 0; i_ < 4; ++i_) atomicAdd(P.ctl + 110 + (ANY ? 16 : 0) + i_, sameQuad[i_]); \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
#define RT_PROBE_QUAD_REPORT()                                                                                                   \
    do {                                                                                                                         \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                                                                 \
            for (int k_ = 0; k_ < 2; ++k_) {                                                                                     \
                const uint32_t* c_ = P.ctl + 100 + 16 * k_;                                                                      \
                printf("top-stats %s: lane-steps %u, of them in quad levels <=0..3: %u %u %u %u; wave steps %u, of them with " \
                       "ALL lanes in levels <=0..3: %u %u %u %u; steps whose lanes sit in 1 / 2 / 3-4 / more distinct quads: %u %u %u %u\n", \
                       k_ ? "shadow" : "closest", c_[0], c_[1], c_[2], c_[3], c_[4], c_[5], c_[6], c_[7], c_[8], c_[9], c_[10], c_[11], c_[12], \
                       c_[13]);                                                                                                  \
            }                                                                                                                    \
    } while (0)
#else
#define RT_PROBE_QUAD_UNITS(r) RT_REF_PAIR_UNITS(r)
#define RT_PROBE_QUAD_VOTE() \
    do {                     \
    } while (0)
#define RT_PROBE_QUAD_STEP() \
    do {                     \
    } while (0)
#define RT_PROBE_QUAD_FLUSH() \
    do {                      \
    } while (0)
#define RT_PROBE_QUAD_REPORT() \
    do {                       \
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
