// sga_kernels.h -- kernel argument blocks and host-callable launchers shared between the
// kernel translation units (*.hip) and the C-ABI engine (sga_engine.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sga {

// The instantiation the last sweep launch of this thread ran (measurement: bench.py writes it beside
// the roofline figure, so that a profile can be matched to the timed kernel).
void note_sweep_kernel(const char *fmt, ...);
const char *last_sweep_kernel();

// Raise a kernel's dynamic-LDS limit once per (device, kernel, size), not on every launch.
hipError_t ensure_lds_limit(const void *kernel, size_t lds_bytes);

constexpr int MAX_CPW = 10;       // coupling-row chunks a wave holds per buffer (dense)
constexpr int T2_MAX_CPW = 4;     // same for the bit-plane form (a chunk is two planes = 8 VGPRs)
constexpr int MAX_WAVES = 16;     // waves per replica workgroup
constexpr int CSR_WAVES_PER_BLOCK = 4;

// One launch of the sweep kernels covers sweeps [0, n_sweeps) of a call; trace / replay /
// schedule pointers are already offset to the launch's first sweep by the host.
struct SweepArgs {
    // couplings
    const void *J;           // dense: [n][ld] of float | int8, zero padded rows
    const int32_t *rowptr;   // CSR, layout entries < 2^31 (null otherwise)
    const long long *rowptr64;  // CSR, always present
    const uint32_t *cvp;     // CSR, slotted layout, packed entries (24-bit column | int8 value << 24) or null
    const int4 *rowinfo;     // CSR, slotted layout (rows padded to whole 64-entry slots), per row:
                             // first slot, slot count, slots from there to an all-zero slot, h (bits)
                             // in slots; the wide sweep forms address rows by it (null otherwise)
    const int2 *cv;          // CSR entries, (column, value bits) interleaved: one 8-byte load each
    const float *h;          // [n]
    const float *diag;       // [n] J_ii (ARITH_F32 only)
    // replica state
    int8_t *spins;           // [R][sstride], +-1, pad = 0
    double *energy;          // [R]
    double *best_energy;     // [R]
    int8_t *best_spins;      // [R][sstride]
    unsigned long long *n_accepted;  // [R]
    // temperatures
    const double *rep_temp;  // [R]
    const double *sched;     // optional T(k, r) = sched[k*sched_ss + r*sched_rs]
    long long sched_ss, sched_rs;
    // replay (parity tests)
    const int32_t *replay_site;  // [R][replay_stride], offset to this launch's first update
    const float *replay_u;
    long long replay_stride;
    // optional outputs
    double *energy_trace;        // [n_sweeps][R]
    uint8_t *accept_trace;       // [R][replay_stride]
    double *dE_trace;            // [R][replay_stride]
    long long ld;                // dense: spins per replica in LDS / HBM (= W * CPW * elems per chunk)
    long long ldj;               // dense: row stride of J in elements (n rounded up to 128 bytes)
    long long plane_bytes;       // bit-plane form: byte offset of the non-zero plane from the sign plane
    long long plane_row_bytes;   // bit-plane form: bytes of one row of one plane = round_up(n, 128) / 8
    int n, sstride, R, n_sweeps;
    int site_mode, arith, rule;
    // many-model batches (dense): replica r belongs to model (replica0 + r) / reps_per_model;
    // J / h / diag of model m start at m * model_stride_j / m * n elements (0 = one model)
    int reps_per_model;
    long long model_stride_j;
    int no_best;  // 1: leave best tracking to the host-driven pass (asymmetric / diagonal J)
    int table_m;  // > 0: J, h integer valued with max_i(sum_j |J_ij| + |h_i|) = table_m
    int table_scale;  // CSR: 1 | 2 -- table entry q stands for dE = 2 q / table_scale (2: J integer, h half-integer)
    int big;      // CSR: spins held as bits in LDS; 1 = one replica per workgroup with 64-bit row
                  // extents (n > ~160k, nnz >= 2^31, long rows), 2 = narrow form, several replicas per
                  // workgroup (short rows)
    int csr_acc;         // CSR: one of CSR_ACC_* (how a row sum is formed)
    int csr_head;        // CSR wide bit forms: head slots per wave the longest row needs (0: eight)
    int csr_pair_ahead;  // CSR narrow table form, every row <= 64 entries: 1 | 2 = the two updates of a pair reduced together
                         // (opt-in), 4 | 8 = that many updates per step, one per row of 16 | 8 lanes (sweep_csr_rows.hip)
    int csr_row_cap;     // CSR: entries of the longest row when that is <= 64 (else 0): sweep_csr_rows.hip picks its build by it
    int table_covers;    // CSR accept-table forms: 1 = no move of this problem lies beyond the table (scale * max_i(sum_j |J_ij| + |h_i|) <= table_m)
    int look_ahead;      // dense, integer problems: reduce LOOK updates together (sweep_dense_impl.h)
    int force_general;   // 1: the general kernel builds even for production arguments (engine option "force_general")
    int tsp_parallel;    // implicit TSP form: updates per step (-1: by the number of cities, 0 | 1: one at a time)
    // A launch over a SUBSET of the replicas (per-replica routing of SGA_FIELD_CACHE_AUTO): workgroup b works on
    // local replica rep_list[b], the grid covers rep_count of them (null: all R, workgroup b = replica b)
    const int *rep_list;
    int rep_count;
    const void *J_aux;   // bit-plane form: the int8 copy [n][ld] (single couplings for the look-ahead)
    // cached-local-field sweep (sweep_clf_impl.h): resident fields F = field_scale * (J s + h)
    void *fields;        // [R][ldf] int16 | int32
    long long ldf;
    int field_bits;      // 16 | 32
    int field_scale;     // 1 | 2 (J integer, h a multiple of 1/2)
    int clf_batched;     // 1: several accepts per round (sweep_clfb_impl.h) where the arguments are the production ones
    int clf_jmax;        // max |J_ij| (integer): the most one flip moves another site's field, in units of 2 scale
    // cached-field sweep of CSR problems (sweep_clf_csr.hip): fields = D [R][ldf] int16, D_i = sum_j J_ij s_j
    const int *clf_hq;   // [n] table_scale * h_i as integers (the static part of the local field)
    int clf_row_max;     // entries of the longest row of the layout (slot padding included)
    uint32_t seed_lo, seed_hi, sweep0, replica0;
};

// TSP-structured couplings that are never stored (sweep_tsp.hip): scaled distance tables + penalties
struct TspArgs {
    const float *nd4;    // [n_cities][npad]: -d[c][c']/4, zero diagonal, zero padded
    const float *nd4t;   // [n_cities][npad]: -d[c'][c]/4 as row c
    int n_cities, npad;  // npad = 256 * waves * passes >= n_cities (LDS column stride in bits, too)
    unsigned int div_magic;  // site / n_cities == (site * div_magic) >> 32 for every site (host verified)
    unsigned int row_bytes;  // 4 * npad
    float a2, b2;        // -A/2 (one position per city), -B/2 (one city per position)
    int f64;             // 0: integer instance, fp32 accumulation is exact; 1: fp64 accumulation
};
size_t tsp_lds_bytes(int n_cities, int npad);

// Wolff cluster rule (sweep_wolff.hip): recorded uniforms for the parity tests (null: Philox)
struct WolffArgs {
    const float *replay_u;   // [R][capacity], consumed in draw order from cursor[r]
    long long capacity;
    long long *cursor;       // [R]
};
size_t wolff_lds_bytes(int n);

struct EnergyArgs {
    int reps_per_model, replica_base;  // many-model batches, as in SweepArgs
    long long model_stride_j;
    const void *J;
    const long long *rowptr;
    const int2 *cv;
    const float *h;
    const int8_t *spins;
    double *energy;
    long long ld, ldj;
    int n, sstride, R;
    // few replicas: the rows of one replica are cut into `slices` contiguous ranges, one workgroup
    // each (grid = R x slices); the (J-part, h-part) sums land in partial[R][slices][2] and
    // launch_energy_finish adds them up in slice order.  slices == 1: one workgroup per replica.
    int slices;
    double *partial;
};
hipError_t launch_energy_finish(const double *partial, int slices, double *energy, int R,
                                hipStream_t st);

// Local fields of all replicas in one pass over the couplings (fields_dense.hip, matrix cores)
struct FieldsArgs {
    const void *J;         // dense [n][ldj] int8 | float
    const int8_t *spins;   // [R][sstride]
    void *Y;               // [R][ldy] int32 (int8 J) | float: Y[r][i] = sum_j J[i][j] s[r][j]
    const float *h;        // [n]
    double *energy;        // [R] or null (finish pass)
    void *fields;          // [R][ldf] int16 | int32 or null (finish pass): field_scale * (Y + h)
    long long ldj, ldy, ldf;
    int n, R, sstride;
    int field_bits, field_scale;
};
// mode 0: int8 J (i8 MFMA) | 1: fp32 J with exact fp32 sums (f32 MFMA) | 2: fp32 J, real valued (f64 MFMA)
hipError_t launch_fields_dense(const FieldsArgs &a, int mode, hipStream_t st);
hipError_t launch_fields_finish(const FieldsArgs &a, bool y_is_int, hipStream_t st);
// cached-local-field sweep: dense integer-valued symmetric problems
hipError_t launch_sweep_clf(const SweepArgs &a, bool j_is_i8, int waves, hipStream_t st);
hipError_t launch_sweep_clfb(const SweepArgs &a, bool j_is_i8, int waves, hipStream_t st);  // several accepts per round
bool sweep_clfb_applies(const SweepArgs &a, bool j_is_i8);
size_t sweep_clfb_lds_bytes(long long ldf, int field_bits, int sstride, int table_m);
int sweep_clf_batch(long long ldj, bool j_is_i8, int waves);
// ... and of CSR problems with integer couplings (rows sorted, |sum_j J_ij s_j| < 2^15)
hipError_t launch_sweep_clf_csr(const SweepArgs &a, int waves, hipStream_t st);
bool sweep_clf_csr_applies(const SweepArgs &a, int waves);
size_t sweep_clf_csr_lds_bytes(long long ldf, int sstride, int table_m);
hipError_t launch_csr_fields_seed(const long long *rowptr, const int2 *cv, const int8_t *spins, int sstride, int n, int R,
                                  short *D, long long ldf, hipStream_t st);
hipError_t launch_scaled_fields(const float *h, int n, int scale, int *hq, hipStream_t st);
size_t sweep_clf_lds_bytes(long long ldf, int field_bits, int sstride, int table_m);
int sweep_clf_waves(long long ldj, bool j_is_i8, int R, int cus, int forced);

// Energies of all replicas of a CSR problem in one pass over the entries (fields_csr.hip)
struct CsrEnergyArgs {
    const long long *rowptr;
    const int2 *cv;
    const unsigned int *sb;  // [n][RW] transposed spin bits
    double *partial;         // [groups][32 RW]
    int n, R, RW, groups;
};
size_t csr_energy_scratch_bytes(int n, int R, int groups);
hipError_t launch_energy_csr_all(const long long *rowptr, const int2 *cv, const float *h, const int8_t *spins, int sstride,
                                 int n, int R, int groups, bool exact32, void *scratch, double *energy, hipStream_t st);

struct ExchangeArgs {
    const double *energies;   // [R_global] by global replica id
    const double *slot_temps; // [R_global]
    int32_t *slot_to_rep;     // [R_global]
    double *rep_temp;         // [R_local]
    long long *attempts, *accepts;  // [R_global] indexed by lower slot of the pair
    const int32_t *start;     // [n_ladders] or null
    const double *u;          // [n_ladders][L/2] or null
    int *n_accepted;          // device counter
    int R_global, R_local, replica0, n_ladders;
    uint32_t seed_lo, seed_hi, round;
    // the ladders this launch decides: all of them (energies by global id), or only those the local replicas fill
    // (whole ladders per rank: `energies` are the local ones, shifted by energy_base = replica0; no gather needed)
    int ladder0, n_ladders_local, energy_base;
};

// launchers (defined in the .hip files); all return hipGetLastError() after the launch
hipError_t launch_sweep_dense(const SweepArgs &a, bool j_is_i8, int acc64, int waves, int cpw,
                              hipStream_t st);  // acc64: 0 fp32 | 1 fp64 any order (exact) | 2 fp64 canonical
// updates reduced together by the production sweeps of an integer problem with this layout (1 =
// one at a time): sweep_dense_impl.h, look-ahead form
int dense_look_ahead(bool t2, bool j_is_i8, bool acc64, int cpw, int waves, int R);
// ternary couplings as two bit-planes (production configuration only)
hipError_t launch_sweep_dense_t2(const SweepArgs &a, int waves, int cpw, hipStream_t st);
// fp32 [n][n] -> sign plane + non-zero plane, each [n][row_bits/32] words, plus nnz[n] (as float)
hipError_t launch_repack_tern2(const float *J, long long ldJ, int n, unsigned int *planes,
                               long long row_bits, float *row_nnz, hipStream_t st);
hipError_t launch_sweep_csr(const SweepArgs &a, int waves_per_replica, hipStream_t st);
bool sweep_csr_rows_applies(const SweepArgs &a);  // sweep_csr_rows.hip: four | eight updates per step
hipError_t launch_sweep_csr_rows(const SweepArgs &a, int waves_per_block, hipStream_t st);
hipError_t launch_sweep_wolff(const SweepArgs &a, const WolffArgs &wa, bool csr, bool j_is_i8, hipStream_t st);
hipError_t launch_sweep_tsp(const SweepArgs &a, const TspArgs &t, int waves, int passes, hipStream_t st);
hipError_t launch_energy_tsp(const EnergyArgs &a, const TspArgs &t, hipStream_t st);
hipError_t launch_fields_tsp(const TspArgs &t, const int8_t *spins, const float *h, const int32_t *sites,
                             int count, double *out, hipStream_t st);
hipError_t launch_tsp_tables(const float *d, long long ldd, int n, int npad, float *nd4, float *nd4t,
                             hipStream_t st);
int csr_waves_per_block(int sstride, int table_m);  // replicas per workgroup that fit LDS (0: none)
bool csr_big_fits(int sstride, int table_m);         // spins as bits: one replica per workgroup
int csr_bits_waves_per_block(int sstride, int table_m);  // spins as bits, narrow form: replicas per workgroup
size_t csr_lds_bytes(int sstride, int table_m, bool bits);  // LDS of one replica (spins + table)
hipError_t launch_energy_dense(const EnergyArgs &a, bool j_is_i8, hipStream_t st);
hipError_t launch_energy_csr(const EnergyArgs &a, hipStream_t st);
hipError_t launch_exchange_neighbor(const ExchangeArgs &a, hipStream_t st);
// ordered list of slot pairs [count][2], one serial chain (exchange_method="all_pairs")
hipError_t launch_exchange_pairs(const ExchangeArgs &a, const int32_t *pairs, int count, hipStream_t st);
hipError_t launch_init_spins(int8_t *spins, int n, int sstride, int R, uint32_t seed_lo,
                             uint32_t seed_hi, uint32_t replica0, hipStream_t st);
// J repack: fp32 [n][ldJ] -> float | int8 [n][ld] zero padded (ld = the packed row stride), plus diag[n]
hipError_t launch_repack_dense(const float *J, long long ldJ, long long rows, int n, void *out,
                               long long ld, bool to_i8, float *diag, hipStream_t st);
// flags[0] = 1 if some J is not an integer in [-127,127]; flags[1] = 1 if some J is outside
// {-1, 0, +1} (ternary couplings can be held as two bit-planes)
// ... flags[5] = 1024 + highest binary exponent of a non-zero value, flags[6] = 1024 - exponent of
// the lowest set bit (0: no non-zero value): is the fp64 sum of a row exact in any order?
hipError_t launch_scan_values(const float *v, long long rows, long long cols, long long ld,
                              int *flags, hipStream_t st);
hipError_t launch_pad_spins(const int8_t *src, int n, int8_t *dst, int sstride, int R,
                            hipStream_t st);
hipError_t launch_unpad_spins(const int8_t *src, int sstride, int8_t *dst, int n, int R,
                              hipStream_t st);
hipError_t launch_gather_diag_csr(const long long *rowptr, const int32_t *colidx, const float *val,
                                  int n, float *diag, hipStream_t st);
// (colidx, val) -> interleaved entries {column, value bits}: a row is one stream of 8-byte loads
// (half the memory instructions, one page instead of two per row)
hipError_t launch_pack_cv(const int32_t *colidx, const float *val, int2 *cv, long long nnz,
                          hipStream_t st);
// row-wise packing into a (possibly padded) layout; cv_src != null: re-pad an interleaved layout
hipError_t launch_pack_cv_rows(const long long *src_ptr, const long long *dst_ptr, const int32_t *colidx,
                               const float *val, const int2 *cv_src, int2 *cv, int n, hipStream_t st);
// CSR row extents between their 32- and 64-bit forms ([n + 1] entries)
hipError_t launch_widen_rowptr(const int32_t *src, long long *dst, long long count, hipStream_t st);
hipError_t launch_rowinfo_fields(int4 *rowinfo, const float *h, int n, hipStream_t st);
hipError_t launch_pack_entries(const int2 *cv, uint32_t *cvp, long long count, int *bad, hipStream_t st);
hipError_t launch_narrow_rowptr(const long long *src, int32_t *dst, long long count, hipStream_t st);
// CSR structure checks on the device.  flags (int[8], zeroed by the caller):
//  [0] rowptr not monotone / not spanning [0, nnz]   [1] column out of range
//  [2] some J or h not an integer                     [3] rows not strictly sorted by column
//  [4] non-zero diagonal entry                        [5] J[i][j] != J[j][i]
//  [6] bits of max_i(sum_j |J_ij| + |h_i|) as float
//  [7] 1024 + highest binary exponent of a non-zero J   [8] 1024 - exponent of the lowest set bit
//  ([2]: bit 0 = some J, bit 1 = some h not an integer)
enum { CSR_BAD_ROWPTR = 0, CSR_BAD_COLUMN, CSR_NOT_INTEGRAL, CSR_UNSORTED, CSR_DIAGONAL,
       CSR_ASYMMETRIC, CSR_ROW_ABS_MAX, CSR_EXP_HI, CSR_EXP_LO, CSR_ROW_J_ABS_MAX, CSR_FLAG_COUNT = 10 };
// how the CSR sweep kernels form a row sum
enum { CSR_ACC_F32_TABLE = 0,   // integer J and h, few distinct uphill moves: fp32 (exact) + accept table
       CSR_ACC_F32 = 1,         // integer J with row sums below 2^24: fp32 accumulation is exact
       CSR_ACC_F64 = 2,         // the fp64 sum of the row's fp32 values is exact (binary exponents of
                                // all J within 53 bits of each other incl. the row length): any order
       CSR_ACC_F64_CANON = 3 }; // anything else: fp64 in the canonical order (sweep_csr.hip)
hipError_t launch_csr_check_rowptr(const long long *rowptr, int n, long long nnz, int *flags,
                                   hipStream_t st);
hipError_t launch_csr_scan(const long long *rowptr, const int32_t *colidx, const float *val,
                           const float *h, int n, int *flags, hipStream_t st);
// sorted = rows strictly sorted by column (binary search); else linear scans of both rows
hipError_t launch_csr_symmetry(const long long *rowptr, const int32_t *colidx, const float *val,
                               int n, bool sorted, int *flags, hipStream_t st);
hipError_t launch_copy_best(const double *energy, const int8_t *spins, double *best_energy,
                            int8_t *best_spins, int sstride, int R, hipStream_t st);

// *out += position-weighted checksum of the 32-bit words of buf (sga_problem_checksum)
hipError_t launch_checksum(const void *buf, long long bytes, unsigned long long *out, hipStream_t st);
// streaming read of `bytes` (a multiple of 16) of device memory, for the bandwidth probe
hipError_t launch_probe_read(const void *buf, long long bytes, float *sink, hipStream_t st);

// single-site operators (IsingModel.get_local_field / flip_spin, SpinDynamics.single_spin_update)
struct PointArgs {
    const void *J;
    const long long *rowptr;
    const int2 *cv;
    const float *h, *diag;
    int8_t *spins;       // the replica's row [sstride]
    double *energy;      // the replica's tracked energy
    unsigned long long *n_accepted;
    const int32_t *sites;  // [count]
    double *out;           // [count] fields (op 0) | out[0] = dE, out[1] = accepted (ops 1, 2)
    long long ld, ldj;
    int n, count, op;      // op 0: fields, 1: flip, 2: metropolis
    int arith, rule;
    double T;
    float u;
    long long model_offset_j;  // element offset of this replica's model in J; h/diag pre-offset
};
hipError_t launch_point_op(const PointArgs &a, bool csr, bool j_is_i8, hipStream_t st);

// 1 -> out[0] if some J[i][j] != J[j][i] or J[i][i] != 0 (per model block of n rows)
hipError_t launch_check_symmetric(const float *J, long long ldJ, long long rows, int n, int *out,
                                  hipStream_t st);
// best[r] = min(best[r], energy[r]) with the spins, one workgroup per replica
hipError_t launch_update_best(const double *energy, const int8_t *spins, double *best_energy,
                              int8_t *best_spins, int sstride, int R, hipStream_t st);

// operator-form PT exchange (cuda_kernels.py:415-443): decide sequentially, then permute rows
hipError_t launch_op_exchange(float *spins, float *tmp_rows, float *energies, const float *temps,
                              const float *u, int32_t *src_of_pos, int *n_accepted,
                              uint32_t seed_lo, uint32_t seed_hi, uint32_t round, int R, int n,
                              hipStream_t st);

// dynamic LDS bytes the dense sweep kernel needs for a given geometry
size_t sweep_dense_lds_bytes(long long ld, int table_m, bool acc64);
// integrality and max_i(sum_j |J_ij| + |h_i|) of a dense problem: out[0] = float bits of the
// max, out[1] bit 0 = some J is not an integer, bit 1 = some h is not an integer, bit 2 = some h is not a
// multiple of 1/2
hipError_t launch_dense_row_nnz(const float *J, long long ld, int n, int *nnz, hipStream_t st);  // sparse matrices given dense
hipError_t launch_dense_to_csr(const float *J, long long ld, int n, const int *rowptr, int *col, float *val, hipStream_t st);
hipError_t launch_dense_row_abs_max(const float *J, long long ldJ, const float *h, long long rows,
                                    int n, unsigned int *out, hipStream_t st);

}  // namespace sga
