// bhs_lab.hip.h -- every compile-time switch of the device code in one place.
//
// A product build (BHS_LAB undefined or 0: what the Makefile builds) takes none of them from the command line: the values
// below are the shipped configuration.  Measurement builds (tools/build_variants.sh: -DBHS_LAB=1 plus -D<switch>=<value>)
// vary one at a time; what each variant measured is in DESIGN.md section 5 and profiles/.
#pragma once
#ifndef BHS_LAB
#define BHS_LAB 0
#endif
#if !BHS_LAB
#undef BHS_PHASES
#undef BHS_PHASES_SPA
#undef BHS_PHASES_CLS
#undef BHS_FILL_ROUNDS
#undef BHS_FILL_NOSTATS
#undef BHS_SPA_U
#undef BHS_WPB
#undef BHS_XCD_CHUNK
#undef BHS_NT_STORES
#undef BHS_GEN_NT
#undef BHS_CLS_COLOUR
#undef BHS_HEAD_PIECE
#undef BHS_DEFER_MUL
#undef BHS_UNIFORM
#undef BHS_WAVE_ATTR
#undef BHS_CAS_ONLY
#undef BHS_MAXB_SYM
#undef BHS_MAXB_NUM
#undef BHS_MAXB_LONG
#undef BHS_LONG_WAVES
#undef BHS_NUM_WAVES
#undef BHS_SYM_WAVES
#undef BHS_LANE_S
#undef BHS_CLS_PARTS
#undef BHS_CLS_STORE_SC1
#undef BHS_CLS_RUN
#undef BHS_CLS_SUPER
#undef BHS_CLS_LAB
#undef BHS_HUB_PRECHECK
#undef BHS_SPA_BLOCK
#endif

#ifndef BHS_PHASES      // per-phase shader-clock accounting of the numeric wave kernel (tools/phase_profile.py)
#define BHS_PHASES 0
#endif
#ifndef BHS_PHASES_SPA      // ... of the bitmap kernel (tools/phase_profile_spa.py)
#define BHS_PHASES_SPA 0
#endif
#ifndef BHS_PHASES_CLS      // ... of the ring kernel (tools/phase_profile_cls.py)
#define BHS_PHASES_CLS 0
#endif
#ifndef BHS_FILL_ROUNDS      // k_fill_queues: rows per thread per reservation
#define BHS_FILL_ROUNDS 16
#endif
#ifndef BHS_FILL_NOSTATS      // k_fill_queues without the per-bin sums
#define BHS_FILL_NOSTATS 0
#endif
#ifndef BHS_SPA_U      // k_row_spa: products per lane per batch
#define BHS_SPA_U 4
#endif
#ifndef BHS_WPB      // wave kernels: waves (rows in flight) per workgroup
#define BHS_WPB 1
#endif
#ifndef BHS_XCD_CHUNK      // wave kernels: rows per XCD-contiguous chunk of the queue
#define BHS_XCD_CHUNK 2048
#endif
#ifndef BHS_NT_STORES      // non-temporal stores of C in the wave kernels (measured: no gain)
#define BHS_NT_STORES 0
#endif
#ifndef BHS_CLS_COLOUR      // k_class_patterns: the slab's 16-byte units coloured over the LDS bank groups (measured: bank conflicts -36 %, LDS cycles -13.5 %, numeric_class no faster, class_patterns 0.05 -> 0.17 ms: off)
#define BHS_CLS_COLOUR 0
#endif
#ifndef BHS_HEAD_PIECE      // classifier: consecutive rows a wave walks (its first goes through the class table whatever it looks like)
#define BHS_HEAD_PIECE 256
#endif
#ifndef BHS_GEN_NT      // general pipeline (lane, quad, wave kernels): non-temporal stores of C
#define BHS_GEN_NT 0
#endif
#ifndef BHS_DEFER_MUL      // numeric wave kernel: 1 valB and the A value stay in registers until the batch is inserted; 2 the A entry index instead; 0 multiply behind the load
#define BHS_DEFER_MUL 1
#endif
#ifndef BHS_UNIFORM      // wave kernels: uniform rows map product -> A entry with one multiply
#define BHS_UNIFORM 1
#endif
#ifndef BHS_WAVE_ATTR      // numeric wave kernel: waves per SIMD asked of the register allocator (6: spills, 3.80 vs 3.49 ms)
#define BHS_WAVE_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
#ifndef BHS_CAS_ONLY      // hash insert: compare-and-swap only (no read first)
#define BHS_CAS_ONLY 1
#endif
#ifndef BHS_MAXB_SYM      // product batches per window: symbolic wave kernel
#define BHS_MAXB_SYM 12
#endif
#ifndef BHS_MAXB_NUM      // ... numeric wave kernel
#define BHS_MAXB_NUM 5
#endif
#ifndef BHS_MAXB_LONG      // ... workgroup kernels
#define BHS_MAXB_LONG 12
#endif
#ifndef BHS_LONG_WAVES      // workgroup kernels: waves per SIMD asked of the register allocator
#define BHS_LONG_WAVES 4
#endif
#ifndef BHS_SYM_WAVES      // symbolic wave kernel: waves per SIMD asked of the register allocator
#define BHS_SYM_WAVES 5
#endif
#ifndef BHS_NUM_WAVES      // numeric wave kernel: the same
#define BHS_NUM_WAVES 6
#endif
#ifndef BHS_LANE_S      // lane-per-row numeric kernel: LDS staging entries per lane
#define BHS_LANE_S 16
#endif
#ifndef BHS_CLS_PARTS      // round 2's class kernel: parts a row's products are dealt in
#define BHS_CLS_PARTS 1
#endif
#ifndef BHS_CLS_STORE_SC1      // class kernels, stores of C: 0 plain, 1 write-through (sc1; rounds 3-4), 2 non-temporal (nt; round 5: numeric_class 1.54 -> 1.39 ms), 3 sc1 nt (1.80), 4 sc0 sc1
#define BHS_CLS_STORE_SC1 2
#endif
#ifndef BHS_CLS_RUN      // ring kernel: rows per run (metadata granularity)
#define BHS_CLS_RUN 8
#endif
#ifndef BHS_CLS_SUPER      // ring kernel: consecutive rows a wave takes before it moves on
#define BHS_CLS_SUPER 64
#endif
#ifndef BHS_CLS_LAB      // ring kernel, wrong results: 1 no slab loads, 2 no stores of C
#define BHS_CLS_LAB 0
#endif
#ifndef BHS_HUB_PRECHECK      // hub kernels: bits tested before the atomic OR
#define BHS_HUB_PRECHECK 3
#endif
#ifndef BHS_SPA_BLOCK      // k_row_spa: threads per workgroup
#define BHS_SPA_BLOCK 1024
#endif

// ---- the phase timers of the measurement builds (no-ops otherwise)
#if BHS_PHASES || BHS_PHASES_SPA || BHS_PHASES_CLS
__device__ unsigned long long g_phase_cycles[16];
#endif
#if BHS_PHASES_SPA
#define BHS_TICK_SPA(i) do { if (NUM && tid == 0) { const unsigned long long t__ = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], t__ - tSpa); tSpa = t__; } } while (0)
#else
#define BHS_TICK_SPA(i) do { } while (0)
#endif
#if BHS_PHASES
#define BHS_TICK(i) do { if (NUM) { const unsigned long long t__ = __builtin_readcyclecounter(); ph[i] += t__ - tPrev; tPrev = t__; } } while (0)
#else
#define BHS_TICK(i) do { } while (0)
#endif
#if BHS_PHASES_CLS      // measurement builds only (tools/phase_profile_cls.py): wave cycles per phase, summed by lane 0
#define BHS_TICK_CLS(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); ph[i] += t__ - tPh; tPh = t__; } while (0)
#else
#define BHS_TICK_CLS(i) do { } while (0)
#endif
