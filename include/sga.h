/*
 * sga.h -- C ABI of the MI355X (gfx950) digital-annealing engine: the drop-in boundary for
 * the reference's Ising spin-sweep hot path.  Plain C, opaque handle, int status returns,
 * no torch types.  Shared library: spin-glass-anneal-rl_amd/csrc/libsga.so.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *
 *   sga_set_dense / sga_set_csr   IsingModel.couplings / external_fields buffers handed to the
 *                                 kernels: spin_glass_rl/annealing/cuda_kernels.py:228-236
 *                                 (dense fp32 [n,n]); core/ising_model.py:69-84 (dense | COO)
 *   sga_init_replicas             ParallelTempering._initialize_replicas,
 *                                 annealing/parallel_tempering.py:175-189 (copy + reset_to_random)
 *   sga_sweep                     CUDAKernelManager.metropolis_update_optimized,
 *                                 annealing/cuda_kernels.py:228-282 (+ fallback :371-398) and
 *                                 SpinDynamics.sweep, core/spin_dynamics.py:73-94, batched over
 *                                 replicas (ParallelTempering._parallel_sweeps, :191-203)
 *   sga_recompute_energies        CUDAKernelManager.compute_energy_optimized,
 *                                 annealing/cuda_kernels.py:284-324 (+ fallback :400-413);
 *                                 IsingModel.compute_energy, core/ising_model.py:149-174
 *   sga_exchange                  CUDAKernelManager.parallel_tempering_exchange_optimized,
 *                                 annealing/cuda_kernels.py:326-369 (+ fallback :415-443) and
 *                                 ParallelTempering._nearest_neighbor_exchange /
 *                                 _attempt_single_exchange, annealing/parallel_tempering.py:214-258
 *   sga_local_fields              IsingModel.get_local_field, core/ising_model.py:176-185
 *   sga_flip                      IsingModel.flip_spin, core/ising_model.py:125-147
 *   sga_update                    SpinDynamics.single_spin_update / _metropolis_update,
 *                                 core/spin_dynamics.py:61-71,131-152
 *   sga_get_best                  best tracking in GPUAnnealer.anneal, gpu_annealer.py:151-153,
 *                                 and ParallelTempering._find_best_solution, :303-313
 *   sga_get_stats                 SpinDynamics.n_accepted / n_rejected, spin_dynamics.py:44-45
 *
 * Pointers: every user buffer may be a host pointer or a device pointer on the engine's device
 * (copies use hipMemcpyDefault); PyTorch-ROCm tensors are passed as tensor.data_ptr().  J / h
 * are packed into engine-owned HBM at set time (device buffers are read in place, never copied
 * whole), so the caller's tensors are not borrowed after the call returns.  Calls on one handle must be serialised by the caller; different
 * handles are independent.  No call throws; on failure a negative code is returned and
 * sga_last_error() (thread-local) describes it.
 */
#ifndef SGA_H
#define SGA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sga_engine sga_engine;

/* status codes */
#define SGA_OK 0
#define SGA_ERR_INVALID -1 /* bad argument / state  -> AnnealingError in the Python shim   */
#define SGA_ERR_DEVICE -2  /* no / invalid GPU, HIP failure -> DeviceError                 */
#define SGA_ERR_MEMORY -3  /* allocation failure                                           */
#define SGA_ERR_UNSUPPORTED -4

/* coupling storage in HBM */
#define SGA_J_AUTO 0 /* bit-planes if J is ternary and n >= 4096, int8 if integer in [-127,127], else fp32;
                        * a sparse integer matrix (n >= 4096, <= 256 non-zeros per row) is kept as CSR when the
                        * field cache is OFF, or AUTO on a problem the cached-field sweep cannot serve */
#define SGA_J_F32 1
#define SGA_J_I8 2
#define SGA_J_T2 3 /* J in {-1,0,+1} as two bit-planes (sign, non-zero): 2 bits per coupling */

/* site order of a sweep */
#define SGA_SITE_RANDOM 0     /* uniform with replacement, Philox4x32-10 (spin_dynamics.py:69) */
#define SGA_SITE_SEQUENTIAL 1 /* i = 0..n-1 (cuda_kernels.py:381)                             */
#define SGA_SITE_REPLAY 2     /* caller-supplied sites (parity tests)                         */

/* arithmetic of the accept rule */
#define SGA_ARITH_F64 0 /* spin_dynamics.py:131-152 (double dE, fp32 exp)                     */
#define SGA_ARITH_F32 1 /* cuda_kernels.py:383-390 (fp32 throughout, minus J_ii s_i)          */

/* single-site update rule (core/spin_dynamics.py:11-16) used by sga_sweep / sga_update */
#define SGA_RULE_METROPOLIS 0 /* spin_dynamics.py:131-152                                      */
#define SGA_RULE_GLAUBER 1    /* spin_dynamics.py:154-171: s_i = +1 w.p. 1/(1+exp(-2 f/T))     */
#define SGA_RULE_HEAT_BATH 2  /* spin_dynamics.py:173-191: same with beta = 1/T formed first   */
#define SGA_RULE_WOLFF 3      /* spin_dynamics.py:193-255: cluster moves (sga_sweep only)         */

/* ---- lifetime ------------------------------------------------------------------------- */
int sga_create(int device, sga_engine **out);
void sga_destroy(sga_engine *e);
const char *sga_last_error(void);
int sga_version(void);
/* Run on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = the
 * engine's own stream (default). */
int sga_set_stream(sga_engine *e, void *hip_stream);

/* ---- problem -------------------------------------------------------------------------- */
/* Dense couplings J[n][ldJ] (fp32, row-major, symmetric as the reference stores them) and
 * fields h[n]. */
int sga_set_dense(sga_engine *e, const float *J, int64_t ldJ, const float *h, int n,
                  int storage);
/* A batch of n_models independent dense problems of the same size (the reference's
 * BatchProcessor workload, annealing/batch_processor.py:231-288): J is the models' matrices
 * stacked row-wise, [n_models*n][ldJ]; h is [n_models][n].  Replicas are split evenly over
 * the models (global replica g belongs to model g / (R_global / n_models)); with a ladder,
 * use n_ladders = n_models so that exchanges stay inside a model. */
int sga_set_dense_batch(sga_engine *e, const float *J, int64_t ldJ, const float *h, int n,
                        int n_models, int storage);
/* CSR couplings (both triangles present), rowptr[n+1], colidx[nnz], val[nnz], h[n]; host or
 * device pointers.  The structure is checked on the device (SGA_ERR_INVALID for extents that are
 * not monotone / do not span [0, nnz], or a column outside [0, n)); rows need not be sorted and
 * duplicates add up.  The arrays are read where they lie (device) or through a staging copy (host)
 * and packed into the engine's layout -- (column, value) entries interleaved, rows of long-row
 * problems padded to whole 64-entry slots; the caller's buffers are not referenced after the call.
 * The same pass classifies how a row sum can be formed exactly (integer / fp64-exact / canonical
 * order, see sga_describe's "path=" and DESIGN.md 2).  Problems whose longest row has <= 64 entries
 * (lattices, low-degree graphs, BASELINE configs[2]) are swept several updates per step -- one per
 * row of 8 or 16 lanes, replayed one at a time when an accepted update touches a later one of the
 * step: the same chain, 3-6 x the one-update rate (sga_describe: "updates_per_step="). */
int sga_set_csr(sga_engine *e, const int32_t *rowptr, const int32_t *colidx, const float *val,
                const float *h, int n, int64_t nnz);
/* Same with 64-bit row extents, for nnz >= 2^31 (BASELINE config 5 at 1000 cities: n = 10^6,
 * nnz ~ 4e9; the reference's COO `couplings` tensor of core/ising_model.py:69-76 has no such
 * limit).  Problems whose int8 spins do not fit LDS (n > ~160k) keep them as bits, up to
 * n ~ 1.3e6; both cases are picked up automatically by sga_init_replicas. */
int sga_set_csr64(sga_engine *e, const int64_t *rowptr, const int32_t *colidx, const float *val,
                  const float *h, int n, int64_t nnz);

/* TSP-structured couplings that are never stored (BASELINE configs[4], examples/tsp_example.py
 * at 1000 cities: 10^6 spins whose CSR is 32 GB).  The problem is the one problems/routing.py:250-328
 * compiles, spin (c, p) = city c at tour position p, index c * n_cities + p, in the convention of
 * the build's encoders.tsp_csr:
 *     J[(c,p),(c,p')] = -city_visit/2     J[(c,p),(c',p)] = -position_fill/2
 *     J[(c,p),(c',p-1)] = -dist[c'][c]/4  J[(c,p),(c',p+1)] = -dist[c][c']/4   (c' != c, p mod n)
 * dist: fp32 [n_cities][ld] (host or device; its diagonal is ignored), h: [n_cities^2] fields.
 * The sweep reads two 4 n_cities-byte distance rows per update instead of a 32 n_cities-byte
 * coupling row.  Row sums are exact (checked here) and rounded once to fp32, as in the stored
 * forms, so the chain equals sga_set_csr's on the same couplings bit for bit (production sweeps work
 * on up to eight updates at once, one per wave, wherever they are independent); sga_describe says
 * "acc=f64" without "-exact" in the one case where that cannot be guaranteed (distances spanning
 * more than ~40 binary orders of magnitude).  Single-site flip / update are not available on this
 * form (SGA_ERR_UNSUPPORTED); local fields are. */
int sga_set_tsp(sga_engine *e, const float *dist, int64_t ld, int n_cities, float city_visit,
                float position_fill, const float *h);

/* ---- replicas ------------------------------------------------------------------------- */
/* R_local replicas live on this engine; they are replicas [replica0, replica0+R_local) of a
 * global set of R_global (R_global == R_local, replica0 == 0 on one GPU).  s0 == NULL draws
 * the initial spins from the Philox stream (domain 2), else s0 is int8 +-1 [R_local][n].
 * Energies are computed, best = initial, counters zeroed, sweep counter = 0.  A failed call leaves the
 * engine WITHOUT replicas (the previous set is released first): later sweeps / state calls return
 * SGA_ERR_INVALID until replicas are initialised again. */
int sga_init_replicas(sga_engine *e, int R_local, int R_global, int replica0, uint64_t seed,
                      const int8_t *s0);
/* Temperature of each local replica (used when sga_sweep gets no schedule). */
int sga_set_temperatures(sga_engine *e, const double *T /* [R_local] */);
/* Temperature ladder(s) over the GLOBAL replica set: n_ladders ladders of R_global/n_ladders
 * slots, slot_temps[R_global]; slot i initially holds replica i.  Sets local temperatures. */
int sga_set_ladder(sga_engine *e, const double *slot_temps /* [R_global] */, int n_ladders);

/* ---- hot path ------------------------------------------------------------------------- */
/* n_sweeps Metropolis sweeps (n single-spin updates each) of every local replica.
 *   sched: optional temperatures T(k, r) = sched[k*sched_sweep_stride + r*sched_replica_stride]
 *          (k = sweep within this call, r = local replica); NULL = current temperatures.
 *   replay_site [R_local][n_sweeps*n] int32, replay_u [R_local][n_sweeps*n] fp32: SITE_REPLAY
 *          needs both; SITE_SEQUENTIAL takes replay_u if non-NULL (else Philox uniforms).
 *   energy_trace: optional [n_sweeps][R_local] doubles, energy after each sweep.
 *   accept_trace / dE_trace: optional [R_local][n_sweeps*n] per-update records (parity tests).
 */
int sga_sweep(sga_engine *e, int n_sweeps, int site_mode, int arith, const double *sched,
              int64_t sched_sweep_stride, int64_t sched_replica_stride,
              const int32_t *replay_site, const float *replay_u, double *energy_trace,
              uint8_t *accept_trace, double *dE_trace);

/* Single-site operators on local replica r (the reference's per-spin API).
 * sga_local_fields: out[k] = fp32(J[sites[k],:].s) + h[sites[k]] as a double.
 * sga_flip: unconditional flip of `site`; *dE = 2 s_i field; the tracked energy follows.
 * sga_update: one Metropolis update at `site` with uniform u at temperature T. */
int sga_local_fields(sga_engine *e, int r, const int32_t *sites, int count, double *out);
int sga_flip(sga_engine *e, int r, int site, double *dE);
int sga_update(sga_engine *e, int r, int site, double T, float u, int arith, int *accepted,
               double *dE);

/* Rule applied by later sga_sweep / sga_update calls (default METROPOLIS).  GLAUBER and
 * HEAT_BATH always consume the uniform and require SGA_ARITH_F64; the per-update dE record of
 * HEAT_BATH is minus the energy change, as the reference returns it (spin_dynamics.py:188). */
int sga_set_update_rule(sga_engine *e, int rule);
/* SGA_RULE_WOLFF (SpinDynamics._wolff_cluster_dense, core/spin_dynamics.py:210-255; for CSR couplings
 * the same rule over a row's stored entries): a sweep is n cluster moves from drawn start sites,
 * always accepted, n_accepted grows by the cluster sizes, the per-update dE record is
 * compute_energy() after minus before, and energies are evaluated from scratch after every sweep
 * (spin_dynamics.py:87).  Needs SGA_ARITH_F64 and n <= ~31 000 (spins, cluster bitmap and queue in
 * LDS), and CSR rows strictly sorted by column (every stored entry is one bond: duplicates, which the
 * single-site rules add up, are refused with SGA_ERR_UNSUPPORTED); not available for sga_update or
 * sga_set_tsp problems.  One uniform per candidate bond:
 * Philox (domain 3) or, for the parity tests, the recorded stream given here -- u [R_local][capacity]
 * fp32, consumed in draw order by the following Wolff sweeps (NULL: back to Philox). */
int sga_set_wolff_replay(sga_engine *e, const float *u, int64_t capacity);

/* Recompute every local replica's energy from scratch: -0.5 s.(J s) - h.s.  Batches of replicas share
 * one pass over the couplings (EnergyComputer.compute_batch_energies, core/energy_computer.py:142-158):
 * dense problems with >= 32 replicas on the matrix cores, CSR problems with >= 64 replicas through a
 * transposed spin-bit matrix; fewer replicas take one pass each.  Same values either way (integer
 * problems: exact; real-valued: the fp64 summation order differs, the fp32-rounded energy does not beyond
 * its last bit). */
int sga_recompute_energies(sga_engine *e);

/* One nearest-neighbour replica-exchange round over the ladder(s)
 * (parallel_tempering.py:214-258: even/odd pairs, accept min(1, exp((b_j-b_i)(E_j-E_i)))).
 *   energies_global: [R_global] doubles indexed by GLOBAL replica id (after an all-gather);
 *                    NULL = the engine's own energies: needs R_local == R_global, or every ladder whole
 *                    on one rank (replica0 and R_local multiples of the ladder length) -- then only the
 *                    local ladders are decided (same Philox keys as in the unsharded run: global ladder
 *                    index), the slot map / exchange statistics of the other ranks' ladders stay
 *                    untouched here, and *n_accepted counts the local ladders' swaps.
 *   start: [n_ladders] int32 0/1 parity per ladder, NULL = Philox (domain 1).
 *   u: [n_ladders][slots/2] doubles in attempt order, NULL = Philox.
 *   n_accepted: optional out (host int). */
int sga_exchange(sga_engine *e, const double *energies_global, const int32_t *start,
                 const double *u, int *n_accepted);

/* Exchange attempts over an ordered list of slot pairs: the CPU branch of the reference's
 * exchange_method="all_pairs" (annealing/parallel_tempering.py:222-232 -- for i < j, gated by
 * `np.random.rand() < 0.1`, _attempt_single_exchange(i, j), :234-258; statistics under
 * min(i, j)).  Each attempt sees the swaps before it.  pairs: host [count][2] int32 global
 * slot indices in attempt order (the gate is the caller's: it only selects the list);
 * u: [count] doubles or NULL (Philox, domain 1); energies_global as in sga_exchange. */
int sga_exchange_pairs(sga_engine *e, const double *energies_global, const int32_t *pairs,
                       const double *u, int count, int *n_accepted);

/* Stateless operator form of CUDAKernelManager.parallel_tempering_exchange_optimized
 * (annealing/cuda_kernels.py:326-369, fallback :415-443): sequential adjacent pairs, fp32,
 * p = exp((1/T[i+1] - 1/T[i]) * (E[i] - E[i+1])), accepted pairs swap spin rows and energies
 * in place.  spins fp32 [R][n], energies fp32 [R], temps fp32 [R]; u fp32 [R-1] or NULL
 * (Philox, domain 1).  Device or host pointers. */
int sga_op_pt_exchange(int device, float *spins, float *energies, const float *temps,
                       const float *u, uint64_t seed, uint32_t round, int R, int n,
                       int *n_accepted);

/* ---- state access --------------------------------------------------------------------- */
int sga_get_energies(sga_engine *e, double *out /* [R_local] */);
/* Stream-ordered form for a DEVICE destination: the copy is enqueued on the engine's stream and the
 * call returns at once -- for consumers ordered behind that stream (sga_set_stream with the caller's
 * stream: an RCCL all-gather of the energies followed by sga_exchange needs no host synchronisation). */
int sga_get_energies_async(sga_engine *e, double *out_device /* [R_local] */);
int sga_get_temperatures(sga_engine *e, double *out /* [R_local] */);
int sga_get_spins(sga_engine *e, int r, int8_t *out /* [n]; r<0: all, [R_local][n] */);
int sga_set_spins(sga_engine *e, int r, const int8_t *s /* recomputes that energy */);
/* r >= 0: best (energy, spins) seen by local replica r at sweep ends; r < 0: the best over
 * all local replicas; *r_out (optional) receives its local index. */
int sga_get_best(sga_engine *e, int r, double *energy, int8_t *spins /* [n] or NULL */,
                 int *r_out);
/* Forget the bests: best = current state (ParallelTempering checks only on record sweeps). */
int sga_reset_best(sga_engine *e);
int sga_get_stats(sga_engine *e, int64_t *accepted /* [R_local] */,
                  int64_t *attempted /* [R_local] */);
int sga_get_slot_map(sga_engine *e, int32_t *slot_to_rep /* [R_global] */);
int sga_get_exchange_stats(sga_engine *e, int64_t *attempts /* [R_global] */,
                           int64_t *accepts /* [R_global] */);
/* What a tempering host loop reads at an event -- energies [R_local], acceptance counters
 * [R_local], the ladder permutation [R_global] (any may be NULL) -- with ONE synchronisation
 * (ParallelTempering._record_statistics / acceptance bookkeeping, parallel_tempering.py:295-301). */
int sga_snapshot(sga_engine *e, double *energies, int64_t *accepted, int32_t *slot_to_rep);
/* Philox key of all later draws (sweeps, exchanges); the counters are unchanged. */
int sga_set_seed(sga_engine *e, uint64_t seed);
int sga_get_sweep_counter(sga_engine *e, uint32_t *sweeps_done, uint32_t *exchange_rounds);
int sga_set_sweep_counter(sga_engine *e, uint32_t sweeps_done, uint32_t exchange_rounds);

/* ---- checkpoint / resume ---------------------------------------------------------------- */
/* Everything that makes a run continue bit-exactly -- spins, tracked and best energies, best
 * configurations, temperatures, ladder permutation and statistics, acceptance counters, seed
 * and stream counters -- as one host blob.  sga_export_state with buf == NULL only reports the
 * size.  Import needs an engine holding the same problem with replicas (and ladder, if one was
 * exported) initialised to the same counts.  (The reference can only save final results:
 * AnnealingResult.save, annealing/result.py:147-165.) */
int sga_export_state(sga_engine *e, void *buf, uint64_t capacity, uint64_t *needed);
int sga_import_state(sga_engine *e, const void *buf, uint64_t size);

/* ---- measurement ---------------------------------------------------------------------- */
/* With timing enabled every sweep-kernel launch is bracketed by HIP events on the launch
 * stream; sga_get_kernel_time returns the launches and their summed duration since the
 * last reset (synchronises). */
int sga_enable_timing(sga_engine *e, int on);
int sga_get_kernel_time(sga_engine *e, int64_t *n_launches, double *total_ms, int reset);
/* Describes the launch geometry chosen for the current problem (for DESIGN/bench output):
 * writes a NUL-terminated string into buf. */
int sga_describe(sga_engine *e, char *buf, int buflen);
/* 64-bit checksum of the problem as the sweep kernels read it (packed couplings / entry layout /
 * distance tables, and h).  Ranks of a sharded run compare it after set-up: every rank must hold the
 * same couplings (the reference replicates the model per device, annealing/multi_gpu.py:110-132). */
int sga_problem_checksum(sga_engine *e, uint64_t *out);
/* Launch geometry of the dense sweep kernels (sga_set_tuning / sga_autotune / heuristic). */
int sga_get_geometry(sga_engine *e, int *waves_per_replica, int *chunks_per_wave);
/* Measurement aid: the kernel instantiation (template arguments, waves per replica) that the calling
 * thread's last sga_sweep launched -- what a rocprofv3 trace of the same command must show. */
int sga_last_kernel(char *buf, int buflen);
/* Measurement aid: GB/s of a plain streaming read (every workgroup its own segment, eight non-temporal 16-byte loads per lane in flight) of a fresh `bytes`-byte
 * device buffer, `reps` passes -- the practical bandwidth of this device beside its spec figure. */
int sga_probe_read_bandwidth(int device, int64_t bytes, int reps, double *gb_per_s);
/* Storage of CSR entries in the one-replica-per-workgroup bit-spin sweep forms (long rows; call before
 * sga_init_replicas).  AUTO: integer-valued problems with |J| <= 127 and n < 2^24 keep a second layout
 * with one dword per entry (24-bit column | 8-bit value) -- half the bytes per row, identical chains;
 * F32: always the (column int32, value fp32) entries; PACKED: fail (SGA_ERR_UNSUPPORTED, at
 * sga_init_replicas) when the packed layout cannot be used.  The choice is latched by
 * sga_init_replicas: a call made while replicas exist takes effect at the next sga_init_replicas
 * (sga_describe reports what the current replicas run).  (No reference counterpart: the
 * reference stores torch COO / dense fp32, core/ising_model.py:56-63.) */
#define SGA_CSR_STORAGE_AUTO 0
#define SGA_CSR_STORAGE_F32 1
#define SGA_CSR_STORAGE_PACKED 2
int sga_set_csr_storage(sga_engine *e, int storage);
/* How sga_sweep evaluates a proposal.  OFF (default): the reference's way -- the local field of the
 * proposed site is formed from its coupling row (IsingModel.get_local_field, core/ising_model.py:176-185):
 * one row read per proposal.  ON / AUTO: the local fields of every replica stay resident in LDS and a row is
 * read only when a proposal is ACCEPTED, to update them -- the reference's incremental mode,
 * core/energy_computer.py:166-173,262-265.  Same sites, uniforms and accept rule on exact integers: the chain
 * equals OFF's bit for bit.  Served:
 *   dense couplings (sga_set_dense, one model): J integer valued and symmetric with a zero diagonal, h in
 *     multiples of 1/2, max_i(sum_j |J_ij| + |h_i|) < 2^24 (2^23 with half-integer h), n <= ~75 000 (int16
 *     fields; ~37 000 with int32); fields seeded by one pass over J on the matrix cores; any rule but
 *     SGA_RULE_WOLFF, every site mode and arithmetic;
 *   CSR couplings (sga_set_csr, or a sparse matrix sga_set_dense kept as CSR): J integer valued and symmetric
 *     in strictly sorted rows (no duplicate entries), zero diagonal, h in multiples of 1/2,
 *     max_i sum_j |J_ij| < 2^15 (only the dynamic part J s of a field is kept, as int16; h is read beside it),
 *     rows of <= 2048 entries, n <= ~72 000; production sweeps (SGA_SITE_RANDOM, SGA_ARITH_F64, Metropolis, no
 *     per-update records) -- other arguments take OFF's kernels for that call, the same chain.
 * ON: sga_sweep fails with SGA_ERR_UNSUPPORTED where the problem does not qualify.  AUTO: falls back to OFF's
 * kernels there -- and while replicas are hot: it starts on OFF's kernels (on the cached fields where the break-even
 * below is above ~0.3: nothing is known yet, and there the row kernels lose more on a cold ladder than the cached
 * fields on a hot one), reads the per-replica acceptance
 * counters back every 4 ... 16 sweeps (when a sga_sweep call starts; a long production call is walked in pieces of 16 sweeps for that) and
 * then routes EACH replica of a dense problem by its own acceptance
 * (break-even = what an update costs its chain on OFF's kernel over what an accept costs it here: 0.25 on
 * bit-planes, 0.39 on int8 rows at n = 10^4, never on fp32 rows): a ladder with a hot end runs as two concurrent
 * launches over disjoint replica lists (option "replica_routing" = 0, and CSR problems: one launch, decided by the
 * hottest replica).  Cost model: an accepted proposal costs ~1.1 - 1.7 us of its replica's serial chain, a
 * rejected one next to nothing -- 100-450 x OFF in the glassy regime annealing ends in. */
#define SGA_FIELD_CACHE_OFF 0
#define SGA_FIELD_CACHE_ON 1
#define SGA_FIELD_CACHE_AUTO 2
int sga_set_field_cache(sga_engine *e, int mode);
/* Form-selection options of ONE engine -- the A/B switches of the measurements and what the parity tests use to
 * force a kernel form -- by name.  None changes a result: every form walks the same chain.  The
 * environment is read once, in sga_create, for the defaults (variable in brackets); afterwards only these calls
 * count, so two engines of one process can run different forms.  SGA_ERR_INVALID: unknown key, value out of range.
 * When a value takes effect: [set] at the next sga_set_dense / sga_set_csr, [init] at the next sga_init_replicas,
 * [sweep] at the next sga_sweep / sga_recompute_energies.  A [set] option changed while couplings are set, or an
 * [init] option changed while replicas exist, cannot act on them any more: sga_sweep then refuses (SGA_ERR_INVALID,
 * naming the key) until the couplings / replicas are set up again -- it never silently runs the form the old value chose.
 *   "look_ahead"            0 | 1 (default)   dense integer problems: several updates reduced together  [sweep; SGA_NO_LOOK_AHEAD]
 *   "force_general"         0 (default) | 1   general kernel builds even for production arguments       [sweep; SGA_FORCE_GENERAL]
 *   "clf_waves"             0 = measured table (default), 1 ... 16 (capped at 8): waves per replica of the windowed cached-field
 *                           sweep (a value selects that form)                                       [sweep; SGA_CLF_WAVES]
 *   "clf_tail_waves"        0 | 1 (default)   cached-field sweep over dense couplings (ON and AUTO): the per-replica
 *                           acceptance is looked at every 4 ... 16 sweeps (when a sga_sweep call or one of its 16-sweep pieces starts); once the mean is below 0.28 of the
 *                           hottest replica's -- the launch is that replica's chain, the chip idles behind it -- every
 *                           replica runs at eight waves.  Same chain                          [sweep; SGA_NO_CLF_TAIL_WAVES]
 *   "clf_batched"           2 (default) | 1 | 0   cached-field sweep under production arguments commits SEVERAL accepts per
 *                           round: all decisions of a window guessed at once, the guess checked against the couplings
 *                           between the accepting sites, the rows applied two at a time (csrc/sweep_clfb_impl.h).  Same
 *                           chain.  2 = while the hottest replica accepts more than ~1 % of its proposals (where the
 *                           form is ahead), 1 = always, 0 = never                               [sweep; SGA_CLF_BATCHED]
 *   "replica_routing"       0 | 1 (default)   SGA_FIELD_CACHE_AUTO routes each replica by its own acceptance (two
 *                           concurrent launches) instead of the whole launch by the hottest replica    [sweep; SGA_NO_REPLICA_ROUTING]
 *   "batched_energy"        0 = one pass over the couplings per replica, 1 (default) = all replicas in one pass where
 *                           that carries the same bits (not for real-valued couplings that need the canonical
 *                           summation order), 2 = always                                                [sweep; SGA_NO_MFMA_ENERGY]
 *   "fields_scratch_mb"     256 (default): cap in MiB of the scratch of the all-replica field / energy pass over dense
 *                           couplings; larger replica sets go through in tiles of 128-replica blocks    [sweep; SGA_FIELDS_SCRATCH_MB]
 *   "csr_updates_per_step"  -1 = by the longest row (default), 0 = one update at a time, 1 | 2 = pair look-ahead,
 *                           4 | 8 = that many updates per step (rows <= 256 entries)                    [init, sweep; SGA_CSR_PAIR_AHEAD]
 *   "tsp_updates_per_step"  -1 = by the number of cities (default), 0 | 1 = one, 2 | 4 | 8              [sweep; SGA_TSP_PARALLEL]
 *   "sparse_route"          0 | 1 (default)   sga_set_dense keeps a sparse integer matrix as CSR        [set; SGA_NO_SPARSE_ROUTE]
 *   "csr_slots"             0 | 1 (default)   long-row CSR problems padded to 64-entry slots at set time [set; SGA_NO_CSR_SLOTS]
 *   "half_integer_table"    0 | 1 (default)   accept table for integer J with half-integer h            [set; SGA_NO_HALF_TABLE]
 *   "force_csr_acc"         0 (default) ... 3: at least this CSR_ACC class (1 f32, 2 f64, 3 f64 canonical) [set; SGA_FORCE_CSR_ACC]
 *   "force_dense_canonical" 0 (default) | 1   canonical fp64 order for every fp64-accumulated dense problem [set; SGA_FORCE_DENSE_CANON]
 *   "zero_slot_every"       0 = 2^21 (default), k: an all-zero slot inside the slotted layout after every k slots [set; SGA_ZERO_SLOT_EVERY]
 *   "csr_bits"              0 | 1 (default)   bit spins where they keep more replicas LDS resident      [init; SGA_NO_CSR_BITS]
 *   "force_csr_bits"        0 (default) | 1   CSR sweeps with bit spins whatever the size               [init; SGA_FORCE_CSR_BIG]
 * sga_option_name enumerates the keys (index 0, 1, ... until SGA_ERR_INVALID).
 * (No reference counterpart: the reference has one code path, core/spin_dynamics.py:61-152.) */
int sga_set_option(sga_engine *e, const char *key, int64_t value);
int sga_get_option(sga_engine *e, const char *key, int64_t *value);
int sga_option_name(int index, char *buf, int buflen);
/* Tuning override (0 = heuristic): waves per replica and sweeps per launch. */
int sga_set_tuning(sga_engine *e, int waves_per_replica, int sweeps_per_launch);
/* Measured choice of the launch geometry / sweep form: times the sweep kernel for every feasible candidate on the
 * current replicas (a few sweeps each) and keeps the fastest -- dense problems: waves per replica; CSR problems
 * (round 4): waves per replica x several updates per step or one, i.e. every form sga_init_replicas chooses between
 * by thresholds (the state travels through the geometry-independent blob of sga_export_state).  The chain does not
 * depend on the form and the replicas' state, best states, counters and kernel-timing statistics are restored, so
 * results are unaffected.  Candidates within 1 % of the fastest are a tie, which goes to the fewest waves / the simpler
 * form (a fixed preference order: other boxes and later profiles see the same pick).  The pick stays as
 * sga_set_tuning / option "csr_updates_per_step" would have set it (readable through sga_get_geometry /
 * sga_get_option).  No-op for sga_set_tsp problems.
 * (No reference counterpart: the reference has no launch geometry.) */
int sga_autotune(sga_engine *e, double *best_ms_per_sweep);
/* What the last sga_autotune of this engine measured: "candidate=ms per sweep;..." (dense: "<waves>x<chunks per wave>",
 * the first entry "heuristic:..." being the untuned choice; CSR: the kernel instantiation).  Empty before. */
int sga_get_autotune_table(sga_engine *e, char *buf, int buflen);

/* ---- form selection, inspectable without a GPU ---------------------------------------------------------------
 * WHICH kernel form sweeps a problem (never what it computes) is a pure function of the problem's traits -- what the
 * set-time scans found -- the replica count, the tuning and the options: csrc/sga_route.cpp, a translation unit
 * without a device call.  The engine fills a sga_route_query from its own state and asks that function; tests fill
 * one by hand and pin the answer (tests/test_host_logic.py: the five BASELINE configs and the fuzz shapes).
 * (No reference counterpart: the reference has one code path, core/spin_dynamics.py:61-152.) */
#define SGA_ROUTE_DENSE 0
#define SGA_ROUTE_CSR 1
#define SGA_ROUTE_TSP 2
#define SGA_ROUTE_MAX_OPTS 32
typedef struct sga_route_query {
    int32_t kind;         /* SGA_ROUTE_DENSE | SGA_ROUTE_CSR | SGA_ROUTE_TSP (sga_set_tsp) */
    int32_t n;            /* spins */
    int32_t n_models;     /* dense batches (sga_set_dense_batch), else 1 */
    int32_t R_local;      /* replicas on this engine (0: none yet) */
    int32_t cus;          /* compute units of the device (MI355X: 256) */
    int32_t tune_waves;   /* sga_set_tuning waves_per_replica (0 = heuristic) */
    int32_t field_cache;  /* SGA_FIELD_CACHE_* */
    /* what the set-time scans found */
    int32_t storage;      /* dense: SGA_J_F32 | SGA_J_I8 | SGA_J_T2 as resolved; CSR: SGA_CSR_STORAGE_* as requested */
    int32_t acc;          /* dense: 0 fp32 (exact) | 1 fp64, any order | 2 fp64 canonical order;
                             CSR: 0 fp32 + accept table | 1 fp32 | 2 fp64 | 3 fp64 canonical */
    int32_t table_m;      /* entries of the accept table (integer problems), 0 = none */
    int32_t table_scale;  /* 2: half-integer fields, table at twice the resolution */
    int32_t clf_ok;       /* the problem qualifies for the cached-local-field sweep (integer, symmetric, ...) */
    int32_t clf_bits;     /* dense: 16 | 32-bit resident fields */
    int32_t clf_scale;    /* dense: 2 = half-integer fields */
    int32_t from_dense;   /* CSR taken from a sparse matrix handed over dense */
    /* CSR structure */
    int64_t nnz;          /* entries (TSP: 4 (cities - 1) n) */
    int64_t max_row_len;  /* entries of the longest row */
    int64_t layout_entries; /* entries of the layout the kernels read (padding included) */
    int32_t slotted;      /* rows padded to whole 64-entry slots */
    int32_t rowptr32;     /* the layout has < 2^31 entries: 32-bit extents exist */
    int32_t packed_ok;    /* every entry fits the packed form (integer |J| <= 127, n < 2^24) */
    int32_t n_cities;     /* TSP */
    /* as laid out (0 = derive from the rest) */
    int32_t sstride;      /* spin stride of the replicas */
    int32_t reserved_;
    int64_t ldj;          /* dense: row stride of the packed couplings in elements */
    int64_t opt[SGA_ROUTE_MAX_OPTS]; /* option values, index = sga_option_name order */
} sga_route_query;
/* zeroes *q and fills the option defaults (the environment is NOT consulted), cus = 256, n_models = 1 */
int sga_route_query_init(sga_route_query *q);
/* one line naming every decision for q: "dense storage=... waves=... chunks_per_wave=... kernel=..." |
 * "csr form=rows|narrow|narrow-bits|wide-bits|wide-bytes spins=... waves=... updates_per_step=..." | "tsp waves=... passes=..."
 * followed by " cached=..." (what sga_set_field_cache would run).  Pure: no device needed. */
int sga_explain_route(const sga_route_query *q, char *buf, int buflen);
/* the query the engine itself would pose for its current problem / replicas / options */
int sga_get_route_query(sga_engine *e, sga_route_query *out);
/* the instantiation this ENGINE's last sweep launch ran (sga_last_kernel: this thread's last launch of any engine) */
int sga_get_last_kernel(sga_engine *e, char *buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* SGA_H */
