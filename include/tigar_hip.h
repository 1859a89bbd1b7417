/*
 * tigar_hip.h -- C-ABI of libtigar_hip.so: the MI355X (gfx950) implementation of the
 * tIGAr extraction hot path.  Loaded from Python with ctypes (tigar_amd/_lib.py).
 *
 * The reference has no C boundary on this path: its seams are Python method contracts
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).
 *
 * Conventions: every call returns 0 on success, non-zero on failure (message via
 * tg_last_error()); device objects are opaque handles with explicit *_destroy; host
 * arrays are plain pointers + sizes; row pointers / sizes are int64, column indices
 * int32 (INDEX_TYPE='int32', tIGAr/common.py:43), values fp64.  Calls on one device are
 * serialised on the library's stream; the library is not thread-safe.
 */
#ifndef TIGAR_HIP_H
#define TIGAR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tg_csr_s *tg_csr_t;   /* device-resident CSR row block            */
typedef struct tg_vec_s *tg_vec_t;   /* device-resident fp64 vector              */
typedef struct tg_ptap_s *tg_ptap_t; /* symbolic plan of K = M^T A M             */
typedef struct tg_cellplan_s *tg_cellplan_t; /* K = M^T A M on a cell-local FE space: dense blocks per cell */
typedef struct tg_comm_s *tg_comm_t; /* RCCL communicator + z-slab descriptor    */

/* ---- runtime ------------------------------------------------------------------ */
int tg_init(int device);                   /* binds the HIP device, creates the stream */
int tg_shutdown(void);
const char *tg_last_error(void);
int tg_sync(void);                         /* hipStreamSynchronize on the library stream */
/* Second stream.  Every call of the library works on the CURRENT stream (0 after tg_init).
 * tg_stream_set(1) makes a second stream current: what a caller enqueues there (e.g. the FE input of the
 * next sub-slab -- in the reference the FE assembly is dolfin's job, tIGAr/common.py:1206-1220) runs
 * beside the work on stream 0.  tg_stream_wait(a, b): stream a waits for everything enqueued so far on
 * stream b (device side, no host synchronisation); the caller orders the hand-over of objects
 * between the streams with it.  Freed device blocks are re-used across the streams safely (events).
 * tg_sync synchronises the current stream only. */
int tg_stream_set(int stream);
int tg_stream_wait(int waiter, int waited);
int tg_device_info(char *name, int name_len, int *num_cu, int64_t *hbm_bytes);
int tg_mem_info(int64_t *free_bytes, int64_t *total_bytes);
int tg_pool_trim(void);
/* caching allocator statistics: bytes / blocks held free in the pool, blocks handed out */
int tg_pool_stats(int64_t *pooled_bytes, int64_t *pooled_blocks, int64_t *live_blocks);                    /* releases the caching allocator's free blocks */
/* HIP-event timers on the library's stream (bench.py roofline measurement). */
int tg_timer_start(int slot);
int tg_timer_stop(int slot, double *ms);   /* synchronises; elapsed ms since start(slot) */

/* Per-kernel accounting filled by the library while it runs (HIP events on the library's
 * stream, resolved at the next host sync).  slot 0: the SpMV inside tg_krylov_solve; slot 1 (count only): those of
 * them whose halo-free rows were computed while the halo exchange with the neighbour ranks was under way; slot 2 (count
 * only): sliced copies (tg_spmv_sell / Krylov solves) built on the slice classes of an earlier matrix with the same
 * pattern instead of a new classification; slot 3 (count only): x passes of the tensor-pattern PtAP that took the FE
 * matrix' pattern from its certificate (written by tg_kron_sum_csr) instead of verifying every column index; slot 4 (count only): host waits
 * (hipStreamSynchronize / transport calls) the device communicator needed inside halo exchanges and all-reduces -- 0 for
 * RCCL and for the IPC communicator, whose exchanges are enqueued only; slot 5 (count only): final stages of the
 * tensor-pattern PtAP (tg_tensor_zstage, tg_tensor2_ptap) that delivered rows of K -- "the line walks ran"; slot 6: CG /
 * GMRES solves that ran as ONE persistent kernel (small systems, csrc/tg_krylov_small.hip): count, and the time of those kernels
 * (their products are also counted in slot 0, with the kernel time spread over them -- an iteration, not a product);
 * slot 7 (count only): CG solves whose products ran on the half-storage copy of csrc/tg_symgrid.hip; slot 8 (count only):
 * direct solves (tg_lu_solve) that ran as a banded Cholesky factorisation (csrc/tg_chol.hip: symmetric positive definite K). */
enum { TG_PROF_KSP_SPMV = 0, TG_PROF_KSP_OVERLAPPED = 1, TG_PROF_SELL_SHAPE_REUSED = 2, TG_PROF_PTAP_CERTIFIED = 3,
       TG_PROF_COMM_HOST_WAITS = 4, TG_PROF_PTAP_TENSOR_WALKS = 5, TG_PROF_KSP_PERSISTENT = 6, TG_PROF_KSP_SYMGRID = 7,
       TG_PROF_LU_CHOLESKY = 8, TG_PROF_NSLOTS = 9 };
int tg_prof_reset(void);
int tg_prof_get(int slot, double *total_ms, int64_t *count);

/* ---- vectors ------------------------------------------------------------------ */
int tg_vec_create(int64_t n, tg_vec_t *out);               /* zero-initialised */
int tg_vec_create_uninit(int64_t n, tg_vec_t *out);        /* contents undefined: for outputs a kernel overwrites entirely
                                                              (PETSc's VecDuplicate leaves the values unset as well) */
int tg_vec_destroy(tg_vec_t v);
int tg_vec_size(tg_vec_t v, int64_t *n);
int tg_vec_upload(tg_vec_t v, const double *host, int64_t n);
int tg_vec_download(tg_vec_t v, double *host, int64_t n);
int tg_vec_fill(tg_vec_t v, double a);
int tg_vec_copy(tg_vec_t dst, tg_vec_t src);
int tg_vec_copy_range(tg_vec_t dst, int64_t dst_off, tg_vec_t src, int64_t src_off, int64_t n);
int tg_vec_axpy(tg_vec_t y, double a, tg_vec_t x);          /* y += a x */
int tg_vec_dot(tg_vec_t x, tg_vec_t y, double *out);        /* deterministic two-stage */
int tg_vec_pointwise_mult(tg_vec_t w, tg_vec_t x, tg_vec_t y);   /* w = x .* y (VecPointwiseMult) */
/* GenericVector::norm: kind 0 = "l1", 1 = "l2", 2 = "linf" (tIGAr/common.py:1330 norm(MTb)) */
int tg_vec_norm(tg_vec_t x, int kind, double *out);
/* as_backend_type(MTb).vec().setValues(zeroDofs, 0)  -- tIGAr/common.py:1154-1158 */
int tg_vec_zero_entries(tg_vec_t y, const int32_t *dofs, int64_t n);
/* same for a slab-local vector holding global entries [g0, g0 + size(y)) */
int tg_vec_zero_entries_offset(tg_vec_t y, const int32_t *dofs, int64_t n, int64_t g0);
/* separable load b[(a,b,c)] = scale * b0[a]*b1[b]*b2[c] (synthetic input) */
int tg_vec_tensor3(tg_vec_t out, int d, const double *const *b1d, const int64_t *n,
                   double scale, int64_t row0, int64_t row1);

/* ---- CSR objects ---------------------------------------------------------------- */
int tg_csr_from_host(int64_t nrows, int64_t ncols, const int64_t *rowptr,
                     const int32_t *col, const double *val, tg_csr_t *out);
int tg_csr_dims(tg_csr_t m, int64_t *nrows, int64_t *ncols, int64_t *nnz);
int tg_csr_download(tg_csr_t m, int64_t *rowptr, int32_t *col, double *val);
/* rows [r0,r1) only: rowptr_out[r1-r0+1] relative to the first entry; col/val NULL = sizes only (two-call
 * protocol).  PETSc counterpart: MatGetRow on a row sample of MTAM (tIGAr/common.py:1194-1204) [ext]. */
int tg_csr_download_rows(tg_csr_t m, int64_t r0, int64_t r1, int64_t *rowptr_out, int32_t *col, double *val,
                         int64_t cap);
int tg_csr_destroy(tg_csr_t m);
/* explicit M^T (the reference's FORM_MT switch, tIGAr/common.py:84,358-360);
 * deterministic: rows of M^T sorted by FE row index. */
int tg_csr_transpose(tg_csr_t m, tg_csr_t *out);
/* IGA dof permutation (tIGAr/common.py:407-433 applyPermutation, 1583-1665 generatePermutation).
 * tg_partition_mode: mt = transposed extraction pattern (rows = IGA dofs, columns = FE rows); fe_owner[j] (host, one
 * entry per FE row, 0 <= owner < world) = the rank owning FE row j; owner_out[i] (host) = the rank owning the most FE
 * rows of dof i's support, the lowest such rank on a tie (scipy.stats.mode), 0 for an empty row.
 * tg_csr_permute_columns: copy of m with column c renamed new_of_old[c] (host, ncols entries, a permutation), rows
 * re-sorted: MatPermute with identity rows. */
/* Field blocks of a matrix on a mixed space whose dofs are numbered field after field (the FE matrices of
 * EqualOrderSpline(nFields > 1), tIGAr/common.py:1891-1914): tg_csr_block cuts out rows [r0, r1) x columns [c0, c1)
 * (columns renumbered from 0); tg_csr_from_blocks puts nf x nf blocks together (equal row counts along a block row, equal column
 * counts down a block column: the fields of a FieldListSpline differ in size), blocks[i * nf + j] at block
 * row i, block column j.  Used to run the scalar tensor-pattern PtAP block by block. */
/* C = A + B on the union of the two patterns (MatAXPY, DIFFERENT_NONZERO_PATTERN [ext]); ascending columns. */
int tg_csr_add(tg_csr_t a, tg_csr_t b, tg_csr_t *out);
int tg_csr_block(tg_csr_t a, int64_t r0, int64_t r1, int64_t c0, int64_t c1, tg_csr_t *out);
/* A copy of `a` (same shape) without the entries whose column c has keep[c] == 0 (host array, ncols bytes).  Used to split
 * M^T A M by residue classes of the columns when a row of the product exceeds the per-row tables of the general kernels
 * (3-D patches of degree >= 5; tIGAr/common.py:1194-1195 has no degree limit). */
int tg_csr_select_columns(tg_csr_t a, const uint8_t *keep, tg_csr_t *out);
int tg_csr_from_blocks(int nf, const tg_csr_t *blocks, tg_csr_t *out);
/* a = d + r on a cell-local FE space (nodes numbered cell after cell, b per cell): d = the entries of the diagonal b x b cell
 * blocks, r = all others, both with a's shape.  100: some row does not hold all b entries of its own block (a is not "what
 * dolfin assembles on the mesh of disconnected cells" plus extra couplings).  The matrix of demos/kl-shell-svk/reef-knot.py:
 * 455-467 -- a T-spline stiffness matrix with contact terms added by hand, the reason extractMatrix takes any A
 * (tIGAr/common.py:1175) -- splits this way: d goes through the cell-block product (tg_cellplan_ptap), r through the general
 * kernels, the results are added on the union pattern (tg_csr_add). */
int tg_csr_split_cells(tg_csr_t a, int b, tg_csr_t *d_out, tg_csr_t *r_out);
/* out row r = row rows[r] of a (host index array; any selection or order, repetitions allowed), columns untouched */
int tg_csr_gather_rows(tg_csr_t a, const int64_t *rows, int64_t n, tg_csr_t *out);
int tg_partition_mode(tg_csr_t mt, const int32_t *fe_owner, int world, int32_t *owner_out);
int tg_csr_permute_columns(tg_csr_t m, const int32_t *new_of_old, tg_csr_t *out);
/* fallback for arbitrary AbstractScalarBasis plug-ins (seam b-2, tIGAr/common.py:1683-1692):
 * rows fed as (row, col, val) triplets from the host loop of generateM; applies the
 * abs(v) > eps filter of tIGAr/common.py:1569, INSERT semantics (last wins), sorts columns. */
int tg_csr_from_triplets(int64_t nrows, int64_t ncols, int64_t nt, const int64_t *rows,
                         const int32_t *cols, const double *vals, double eps, tg_csr_t *out);

/* ---- extraction-operator build (generateM) ------------------------------------- */
/* One univariate B-spline direction of a tensor-product basis (BSpline1,
 * tIGAr/BSplines.py:164-351) plus the FE node coordinates along that direction. */
typedef struct {
  int32_t p;              /* degree                                              */
  int32_t nknots;         /* len(knots)                                          */
  const double *ghost;    /* ghostKnots, length nknots + 2*(p+1)  (:204-212)     */
  int32_t mult_first;     /* multiplicities[0]                                   */
  int32_t mult_last;      /* multiplicities[-1]                                  */
  int32_t ncp;            /* number of basis functions (:273-277)                */
  int64_t nnodes;         /* FE nodes along this direction                       */
  const double *nodes;    /* their parametric coordinates (host)                 */
} tg_dir_t;

/* Replaces AbstractCoordinateChartSpline.generateM / generateM_control
 * (tIGAr/common.py:1460-1578) + BSpline.getNodesAndEvals (tIGAr/BSplines.py:450-503)
 * + basisFuncsInner (tIGAr/BSplines.py:73-120) for BSpline bases on the implicit
 * tensor FE node grid.  Rows = nodes in lexicographic order (direction 0 fastest),
 * restricted to [row0,row1) (a z-slab); column = i + j*ncp0 + k*ncp0*ncp1 + col_offset;
 * entries with fabs(v) > eps kept; columns sorted.  ncols = total columns of the block. */
int tg_extract_csr_tensor(int d, const tg_dir_t *dirs, int32_t col_offset, int64_t ncols,
                          double eps, int64_t row0, int64_t row1, tg_csr_t *out);
/* The explicit transpose M^T of the same operator (the reference's FORM_MT object,
 * tIGAr/common.py:84,358-360), written directly: rows = spline dofs [dof0,dof1) of this field,
 * columns = fe_row_offset + lexicographic FE node index; bit-identical to transposing
 * tg_extract_csr_tensor's result. */
int tg_extract_csr_tensor_t(int d, const tg_dir_t *dirs, int64_t fe_row_offset, int64_t fe_rows_total,
                            double eps, int64_t dof0, int64_t dof1, tg_csr_t *out);
/* Matrix-free y = M x for the same operator (rows [row0,row1), x holding the columns
 * [x_col0, x_col0+size(x))): prolongation u = M*U (tIGAr/common.py:1259) without M in memory. */
int tg_extract_apply_tensor(int d, const tg_dir_t *dirs, int32_t col_offset, double eps, int64_t row0,
                            int64_t row1, tg_vec_t x, int64_t x_col0, tg_vec_t y);
/* Same with explicit node coordinates x[nrows*d] (dolfin-supplied / DG nodes). */
int tg_extract_csr_points(int d, const tg_dir_t *dirs, int32_t col_offset, int64_t ncols,
                          double eps, const double *x, int64_t nrows, tg_csr_t *out);
/* vertical concatenation of row blocks (multi-field M: one block per field) */
int tg_csr_vstack(int nblocks, const tg_csr_t *blocks, tg_csr_t *out);
/* The same stacked matrix WITHOUT copying entries: a loose-row view whose rows point into the
 * blocks' own col/val arrays (20 B per row are written instead of 12 B per entry).  The blocks
 * (canonical or loose-row) must outlive the view; accepted where loose-row matrices are. */
int tg_csr_vstack_view(int nblocks, const tg_csr_t *blocks, tg_csr_t *out);
/* Incremental vstack: K is assembled slab by slab into ONE allocation (no second copy of a 70 GB
 * matrix): create with an nnz capacity estimate, append row blocks in order (the block's arrays
 * are copied device-to-device; the capacity grows geometrically if the estimate was short),
 * finish to obtain the CSR object. */
typedef struct tg_csr_builder_s *tg_csr_builder_t;
int tg_csr_builder_create(int64_t nrows_total, int64_t ncols, int64_t nnz_capacity, tg_csr_builder_t *out);
int tg_csr_builder_append(tg_csr_builder_t b, tg_csr_t block);
int tg_csr_builder_finish(tg_csr_builder_t b, tg_csr_t *out);   /* destroys the builder */
int tg_csr_builder_destroy(tg_csr_builder_t b);                 /* abandons an unfinished builder */
/* 1-D evaluation only (device twin of BSpline1.getKnotSpan/getNodes/basisFuncs):
 * idx[n*(p+1)], val[n*(p+1)] in the reference's order (span-p .. span). */
int tg_eval_basis_1d(const tg_dir_t *dir, const double *u, int64_t n, int32_t *span,
                     int32_t *idx, double *val);

/* ---- extraction application ------------------------------------------------------ */
/* y = A x   (prolongation u = M*U, tIGAr/common.py:1259; K*p inside the Krylov solve).
 * x must cover columns [col_base, col_base + size(x)). */
int tg_spmv(tg_csr_t a, tg_vec_t x, tg_vec_t y);
/* same with x covering only the columns [x_col0, x_col0 + size(x)) of a row block whose
 * entries all fall in that range (z-slab pieces of M, M^T, K) */
int tg_spmv_offset(tg_csr_t a, tg_vec_t x, int64_t x_col0, tg_vec_t y);
/* Sliced, pattern-compressed copy of the VALUES of `a` for repeated products (tg_sell.hip).  The
 * Krylov solvers build and drop it themselves for every solve (the K p of KSP on PETSc AIJ,
 * tIGAr/common.py:1255-1258); this entry point keeps one on the matrix so that tg_spmv / tg_spmv_offset
 * use it as well.  It is a snapshot: request it again after changing values.  Stencil-like matrices
 * (K = M^T A M of tensor-product patches) are accepted -- rows in slices of 64, one sorted union of
 * column offsets col - row per slice, values as dense [offsets][64] blocks, 8 B instead of 12 B per
 * entry and no gather -- others are declined and keep the CSR kernel (nclasses = 0).  Rows are summed
 * sequentially in ascending column order (PETSc's order); padded positions add +0.0.
 * enable = 0 drops the copy.  nclasses / padded (may be NULL): dictionary size, doubles stored. */
int tg_spmv_sell(tg_csr_t a, int enable, int *nclasses, int64_t *padded);
/* Half-storage product (csrc/tg_symgrid.hip) -- what the CG solve of tg_krylov_solve uses for its K p when K is a
 * SYMMETRIC box stencil of radius 1..3 on a 3-D grid held by one rank (K = M^T A M of one scalar field on a 3-D patch;
 * the reference hands K to PETSc's KSPCG, tIGAr/common.py:1255-1258, whose premise is a symmetric operator): the diagonal
 * and the entries above it are stored once (172 of 343 per row for p = 3) and used for the row and for the transposed
 * entry; the scatter goes through a ring of LDS windows per (x, y) patch walked along z, deterministically (no global
 * atomics).  `a` is the whole matrix (row0 = 0) or the block of rows [row0, row0 + nrows) a rank holds -- whole planes of
 * the grid, all columns: rows in its first planes also take their entries below the block from the CSR arrays (the
 * previous rank stores those as ITS upper triangle), what is scattered beyond the block is dropped; the solve does this
 * per z slab after the usual halo exchange of the vector.  This entry point plans the copy for `a`, checks it against the
 * CSR product on a pseudo-random vector, and, with x (all columns) and y given, computes y = a x with it.  *accepted = 0:
 * `a` has no such structure or is not symmetric (nothing is computed).  value_bytes / staging_bytes (may be NULL): bytes of
 * K one product reads / size of the window staging. */
int tg_spmv_symgrid(tg_csr_t a, int64_t row0, tg_vec_t x, tg_vec_t y, int *accepted, int64_t *value_bytes,
                    int64_t *staging_bytes);
/* Y = A X for k <= 4 right-hand sides (cpFuncs = M_control * P, tIGAr/common.py:367-380);
 * X, Y are column-major host arrays. */
int tg_spmm_host(tg_csr_t a, const double *X, int k, double *Y);
/* multTranspose / extractVector (tIGAr/common.py:97-109,1142-1160): y = M^T b using the
 * explicit transpose mt (scatter-free, deterministic). */
int tg_spmv_t(tg_csr_t mt, tg_vec_t b, tg_vec_t y);

/* extractMatrix (tIGAr/common.py:1176-1204): K = M^T A M (PETSc MatPtAP [ext]).
 * symbolic: pattern + plan; numeric: values (+ optional fused zeroRowsColumns).
 * Row-block form: computes K rows [i0,i1) from mt rows [i0,i1) (local rows of `mt`
 * start at global row mt_row0), A rows starting at a_row0, M rows starting at m_row0. */
int tg_ptap_symbolic(tg_csr_t a, int64_t a_row0, tg_csr_t m, int64_t m_row0, tg_csr_t mt,
                     int64_t mt_row0, tg_ptap_t *plan);
int tg_ptap_numeric(tg_ptap_t plan, tg_csr_t a, tg_csr_t m, tg_csr_t mt,
                    const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out);
int tg_ptap_destroy(tg_ptap_t plan);
/* Which kernels tg_ptap_symbolic plans for: 0 = chosen from the operands (default), 1 = two row-wise Gustavson products
 * with one wave per row (csrc/tg_ptap_wave.hip: faster for long operand rows, 3-D p >= 3), 2 = the fused
 * workgroup-per-row kernel (csrc/tg_ptap.hip).  The result is the same matrix either way (MatPtAP, tIGAr/common.py:1194-1195);
 * a caller that streams the product in row blocks with overlapping operands states its preference here.  Returns the
 * previous setting. */
int tg_ptap_prefer(int kernels);
/* extractMatrix on a CELL-LOCAL FE space (the meshes of disconnected cells of tIGAr/RhinoTSplines.py:195-240 and
 * tIGAr/BSplines.py:800-860: b nodes per cell, numbered cell after cell), where an assembled A is block diagonal with one
 * dense b x b block per cell: K = sum_c S_c^T (M_c^T A_c M_c) S_c -- dense element matrices out of LDS, merged into K by
 * the look-up stage of the wave kernels.  The plan depends on M only (host-built: md [ncell][b][nfmax] dense rows of M per
 * cell over the cell's function list fl [ncell][nfmax] (nf[c] of them used), incidence: row i -> rows c * nfmax + q of
 * the element matrices holding function i, borrowed for the plan's lifetime; max_k / mean_k: row lengths of K).
 * tg_cellplan_ptap returns 100 when `a` is not such a block-diagonal matrix (use tg_ptap_*). */
int tg_cellplan_create(int64_t ncell, int b, int nfmax, int64_t ncols, const double *md_host, const int32_t *fl_host,
                       const int32_t *nf_host, tg_csr_t incidence, int max_k, double mean_k, tg_cellplan_t *out);
int tg_cellplan_ptap(tg_cellplan_t plan, tg_csr_t a, const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out);
int tg_cellplan_destroy(tg_cellplan_t plan);
/* extractMatrix for ARBITRARY sparse operands on a CONNECTED mesh (cells share nodes: what dolfin assembles on a Q_p / P_p mesh,
 * tIGAr/common.py:1194-1195 with an M that is not used as a Kronecker product), csrc/tg_elemsplit.hip.  A = sum_c R_c^T A_c R_c
 * for ANY splitting of its entries over the cells that hold both nodes, hence K = sum_c (R_c M)^T A_c (R_c M): dense products
 * per cell, merged into K by stored places.  Everything runs on the device; the input besides the CSR operands is the cells'
 * node lists (dolfin: V.dofmap().cell_dofs(c)).
 * tg_cells_t: node lists [ncell][b] (b <= 128), from a host array or, for this package's stand-in of the FE space (continuous
 * Q_p on a tensor-product grid of nodes, direction 0 fastest), generated on the device for a box of elements.
 * tg_elemplan_create: the plan for the cells [own0, own1) of `cells` -- listed cells outside that range only take part in the
 * ownership rule (an entry of A belongs to the FIRST listed cell that holds both its nodes), so a mesh can be worked off in
 * chunks whose K are added.  m holds the rows [m_row0, ...) of M (global columns) of all nodes of the own cells; `cells` and `m`
 * are borrowed for the plan's lifetime.  Status 100: the cells do not qualify (more than 128 functions in a cell, a function in
 * more than 128 cells).
 * tg_elemplan_ptap: rows [dof0, dof1) (tg_elemplan_info) of sum over the own cells, global columns, MatZeroRowsColumns applied
 * when zero_dofs is given.  a holds the rows [a_row0, ...) of A; every entry of the rows [check_row0, check_row1) must couple two
 * nodes of a listed cell, else status 100 (a coupling added by hand, demos/kl-shell-svk/reef-knot.py:455-467: take tg_ptap_*).
 * The pattern of K and the places are found on the first product and kept by the plan. */
typedef struct tg_cells_s *tg_cells_t;
typedef struct tg_elemplan_s *tg_elemplan_t;
int tg_cells_from_host(const int32_t *nodes_host, int64_t ncell, int b, tg_cells_t *out);
int tg_cells_from_grid(int d, const int64_t *nodes_per_dir, int degree, const int64_t *elem_lo, const int64_t *elem_hi,
                       tg_cells_t *out);
int tg_cells_dims(tg_cells_t cells, int64_t *ncell, int *b);
int tg_cells_download(tg_cells_t cells, int32_t *nodes_host);
int tg_cells_destroy(tg_cells_t cells);
int tg_elemplan_create(tg_cells_t cells, int64_t own0, int64_t own1, tg_csr_t m, int64_t m_row0, tg_elemplan_t *out);
int tg_elemplan_info(tg_elemplan_t plan, int64_t *dof0, int64_t *dof1, int *nfmax, int *ninc_max, int64_t *k_nnz);
int tg_elemplan_ptap(tg_elemplan_t plan, tg_csr_t a, int64_t a_row0, int64_t check_row0, int64_t check_row1,
                     const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out);
int tg_elemplan_destroy(tg_elemplan_t plan);
/* The same product for a matrix that holds its dense cell blocks PLUS couplings outside them (contact / penalty terms added by
 * hand to a T-spline matrix: demos/kl-shell-svk/reef-knot.py:455-467, the reason extractMatrix takes any A, tIGAr/common.py:
 * 1175): k_out = M^T D M from the blocks D read in place (no copy of A), r_out = the remainder A - D with A's shape, whose
 * product goes through the general kernels and is added on the union pattern (tg_csr_add).  100: some row lacks entries of
 * its own block.  No boundary conditions here (MatZeroRowsColumns follows the sum). */
int tg_cellplan_ptap_extras(tg_cellplan_t plan, tg_csr_t a, tg_csr_t *k_out, tg_csr_t *r_out);
/* The rows of `a` that hold entries, ascending, into rows_host[0 .. min(*count, cap)); cap = 0 asks for the count only. */
int tg_csr_nonempty_rows(tg_csr_t a, int64_t cap, int64_t *rows_host, int64_t *count);

/* K = R^T K_u R for a 0/1 matrix R with ONE entry per row, MatZeroRowsColumns (tIGAr/common.py:1196-1204) fused, on a PLAN:
 * after the tensor line walks ran on the unwrapped space of a patch with periodic directions (tIGAr/BSplines.py:204-212,
 * 310-319: `% ncp` in getNodes; R = which unwrapped function is which spline function) the rows of K_u that R identifies are
 * added and the columns renamed.  tg_foldplan_create takes the pattern of K from a first product made with the general
 * kernels (tg_ptap_numeric(K_u, R, R^T)) and stores the place of every entry of K_u in its row of K; tg_foldplan_apply then
 * is one pass over K_u without any look-up (status 100: K_u has another pattern than the plan's -- checksum of its row
 * pointer and columns -- or a row of K too long for the 16-bit places: use the general kernels).  ku: rows ku_row0 ... of
 * K_u with global (unwrapped) columns; r: all rows of R; rt: the rows rt_row0 ... of R^T = the rows of K (borrowed by the
 * plan: it must outlive it). */
typedef struct tg_foldplan_s *tg_foldplan_t;
int tg_foldplan_create(tg_csr_t ku, int64_t ku_row0, tg_csr_t r, tg_csr_t rt, int64_t rt_row0, tg_csr_t k, tg_foldplan_t *out);
int tg_foldplan_apply(tg_foldplan_t plan, tg_csr_t ku, const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out);
int tg_foldplan_destroy(tg_foldplan_t plan);
/* extractMatrix when the extraction operator is a Kronecker product (tensor B-splines): one
 * contraction stage  out = P^T cur P  with P = (x)_k F_k, F_k = the 1-D matrix of direction k
 * (n x m CSR + its transpose, host pointers) or the identity (rowptr == NULL).  `cur` is an
 * arbitrary sparse row block on the tensor index space dims_in (rows starting at cur_row0,
 * global columns); rows [out_row0,out_row1) of the result are produced.  Accumulates in a dense
 * LDS box addressed directly (no hashing).  Returns 100 when the box would not fit in LDS --
 * use tg_ptap_symbolic/numeric with explicit operators instead. */
typedef struct {
  int64_t n, m;                 /* F is n x m                                   */
  const int32_t *rowptr;        /* n+1, NULL = identity (direction not contracted) */
  const int32_t *col;
  const double *val;
  const int32_t *t_rowptr;      /* transpose, m+1                               */
  const int32_t *t_col;
  const double *t_val;
} tg_kron1d_t;
int tg_ptap_kron(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                 int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                 tg_csr_t *out);
/* Intermediate stage of a direction-by-direction product (no boundary conditions).  The result
 * is a LOOSE-ROW matrix: its rows lie where the kernel reserved space for them (the row-reorder copy
 * is skipped); it may only be passed on to tg_ptap_kron / tg_ptap_kron_stage (as `cur`),
 * tg_csr_vstack (with other loose-row blocks), tg_csr_compact, tg_csr_dims, tg_csr_download and
 * tg_csr_destroy -- every other entry point rejects it. */
int tg_ptap_kron_stage(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in,
                       const tg_kron1d_t *fac, int64_t out_row0, int64_t out_row1, tg_csr_t *out);
/* Last stage with the result appended to a slab-wise builder of K (tg_csr_builder_*): the rows
 * [out_row0,out_row1) must be the next rows the builder expects; saves the block allocation and the
 * copy of tg_csr_builder_append.  Returns 100 like tg_ptap_kron when the kernel declines. */
int tg_ptap_kron_append(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in,
                        const tg_kron1d_t *fac, int64_t out_row0, int64_t out_row1,
                        const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_builder_t dest);
/* ---- tensor-pattern extractMatrix (tIGAr/common.py:1176-1204 for tensor-product patches) ----
 * K = M^T A M in three "line walk" passes when A carries the element-coupling pattern of the Q_p node
 * grid (what dolfin assembles for any form on V); the pattern is verified entry by entry while A is read
 * and status 100 means "another pattern: use tg_ptap_kron* / tg_ptap_*".  No column decode, no LDS, no
 * atomics: results are bit-reproducible.  See csrc/tg_tensor_body.h. */
typedef struct {
  int p;                   /* spline degree = degree of the CG Lagrange grid, 1..3                    */
  int nel;                 /* elements; nfe = p*nel+1 FE nodes, ncp = nel+p spline functions           */
  const double *wl;        /* host [nel][p+1][p+1]: value at node p*e+j of spline function e+q         */
} tg_tensor_dir_t;
typedef struct tg_tensor_plan_s *tg_tensor_plan_t;
typedef struct tg_tensor_planes_s *tg_tensor_planes_t;
int tg_tensor_plan_create(int d, const tg_tensor_dir_t *dirs, tg_tensor_plan_t *out);
/* The same plan for ONE BLOCK of a space whose fields sit on different spline bases over one FE node grid -- the
 * components of a div- or curl-conforming B-spline (tIGAr/compatibleSplines.py:21-66: generateFieldsCompat; the spaces of
 * demos/taylor-green/taylor-green-3d.py:42-43), a FieldListSpline (tIGAr/common.py:1949-1970):
 *     K_fg = M_f^T A_fg M_g   with  M_f != M_g.
 * Per direction: p = degree of the FE grid (all fields of such a space extract to one Q_p grid: BSpline.getDegree is the
 * largest directional degree, tIGAr/BSplines.py:580-588), pr / pc = spline degree of the row / column side (<= p), wlr /
 * wlc = their local weights [nel][p+1][p+1] padded with zeros to p + 1 functions per element.  The passes are those of the
 * square plan; the last one writes the true pattern (row function i x column functions [i - pr, i + pc]) and leaves the
 * padding out.  tg_tensor_zstage takes no zero dofs for such a plan (MatZeroRowsColumns acts on the assembled matrix). */
typedef struct {
  int p, nel;
  int pr, pc;
  const double *wlr, *wlc;
} tg_tensor_pair_dir_t;
int tg_tensor_plan_create_pair(int d, const tg_tensor_pair_dir_t *dirs, tg_tensor_plan_t *out);

int tg_tensor_plan_destroy(tg_tensor_plan_t plan);
/* The same for a patch with TWO parametric directions and nfields fields on one scalar basis (M = I (x) M_y (x) M_x,
 * dofs field after field; degrees 1..4): K = M^T A M of the whole matrix in two passes, MatZeroRowsColumns fused.
 * A must hold nfields^2 blocks that all carry the element-coupling pattern (verified; 100 = another pattern). */
int tg_tensor2_plan_create(int nfields, const tg_tensor_dir_t *dirs, tg_tensor_plan_t *out);
/* The same for ONE block (f, g) of fields on different tensor bases over one 2-D Q_P node grid (2-D compatible B-splines,
 * tIGAr/compatibleSplines.py:21-66, demos/taylor-green/taylor-green-2d.py): dirs[2] as for tg_tensor_plan_create_pair;
 * tg_tensor2_ptap then takes the scalar block A_fg and returns K_fg = M_f^T A_fg M_g (no boundary conditions). */
int tg_tensor2_plan_create_pair(const tg_tensor_pair_dir_t *dirs, tg_tensor_plan_t *out);
int tg_tensor2_ptap(tg_tensor_plan_t plan, tg_csr_t a, const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *out);
/* x and y passes over the FE planes [z0,z1) of the last direction; `a` holds FE rows from a_row0 on (whole
 * planes, global columns).  The result (dense blocks, no indices) feeds tg_tensor_zstage and can be kept
 * across sub-slabs. */
int tg_tensor_planes(tg_tensor_plan_t plan, tg_csr_t a, int64_t a_row0, int z0, int z1, tg_tensor_planes_t *out);
int tg_tensor_planes_destroy(tg_tensor_planes_t p);
/* An FE matrix on the plan's node grid with ANOTHER pattern than the element-coupling one (entries added by hand, or
 * missing): on_pattern = the matrix with exactly that pattern holding A's entries that lie on it (zeros elsewhere; it
 * carries the pattern certificate), remainder = the other entries as a CSR matrix of A's shape.
 * M^T A M = M^T on_pattern M (tg_tensor_planes / tg_tensor_zstage) + M^T remainder M (tg_ptap_*), tg_csr_add.
 * Returns 100 when A is not a square matrix on the whole node grid. */
int tg_tensor_split(tg_tensor_plan_t plan, tg_csr_t a, tg_csr_t *on_pattern, tg_csr_t *remainder);
/* z pass: rows of K for the dof planes [ka,kb) from pieces that together hold the FE planes in their support;
 * MatZeroRowsColumns(zero_dofs, diag) fused; appended to `dest` (next rows of a slab-wise builder) or, with
 * dest == NULL, returned as a new matrix in *out. */
int tg_tensor_zstage(tg_tensor_plan_t plan, int npieces, const tg_tensor_planes_t *pieces, int ka, int kb,
                     const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_builder_t dest, tg_csr_t *out);
/* canonical CSR copy of a loose-row matrix */
int tg_csr_compact(tg_csr_t in, tg_csr_t *out);
int tg_csr_is_loose(tg_csr_t m, int *loose);
int tg_csr_rowptr_at(tg_csr_t m, int64_t r, int64_t *out);   /* rowptr[r], 0 <= r <= nrows */
/* MatZeroRowsColumns(K, zeroDofs, diag) [ext] as called at tIGAr/common.py:1200;
 * K holds global rows [row0, row0+nrows). */
int tg_zero_rows_cols(tg_csr_t k, int64_t row0, const int32_t *dofs, int64_t n, double diag);
/* out = a X + b Y diag(colscale) for X, Y on one sparsity pattern (checked); colscale may be NULL.
 * Tangent matrices J(u) of Newton loops over the path (tIGAr/common.py:1304-1348), e.g.
 * K + M diag(g'(u)): MatAXPY(SAME_NONZERO_PATTERN) + MatDiagonalScale [ext]. */
int tg_csr_combine(double a, tg_csr_t X, double b, tg_csr_t Y, tg_vec_t colscale, tg_csr_t *out);

/* ---- Krylov solve (solveLinearSystem, tIGAr/common.py:1236-1263; seam b-4) -------- */
enum { TG_KSP_CG = 0, TG_KSP_GMRES = 1, TG_KSP_BICGSTAB = 2 };
/* TG_PC_CHEBYSHEV (CG only): `restart` steps of the Chebyshev iteration for D^-1 K on [lmax / ratio, 1.1 lmax] as a fixed
 * polynomial preconditioner (PETSc: PCKSP with KSPCHEBYSHEV + PCJACOBI [ext]; the stand-in for "sor" / "ilu" / "icc" /
 * "bjacobi", which need triangular sweeps); lmax from 20 steps of the power method, capped by the Gershgorin bound. */
enum { TG_PC_NONE = 0, TG_PC_JACOBI = 1, TG_PC_CHEBYSHEV = 2 };
/* status: 0 converged (rtol), 1 converged (atol), -1 max iterations, -2 breakdown/NaN,
 * -3 stagnation (GMRES: 25 restart cycles in a row without progress) */
int tg_krylov_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int method, int pc, double rtol,
                    double atol, int maxit, int restart, tg_comm_t comm, int *iters,
                    double *resnorm, int *status);
/* same with flags: TG_KSP_NONZERO_GUESS = x holds the initial guess (dolfin's solver parameter
 * "nonzero_initial_guess" [ext]; the convergence test stays relative to ||B b||, PETSc's default) */
enum { TG_KSP_NONZERO_GUESS = 1, TG_KSP_STAGNATION_GUARD = 2, TG_KSP_SYMMETRIC = 4 };
/* TG_KSP_SYMMETRIC: the caller vouches that k is symmetric (the premise of KSPCG, which PETSc does not check either; set by
 * ExtractedSpline.assembleMatrix for forms that are symmetric by construction): the half-storage copy of a CG solve is still
 * built from rows that are checked one by one against the box stencil, but not compared with the CSR product. */
int tg_krylov_solve_flags(tg_csr_t k, tg_vec_t b, tg_vec_t x, int method, int pc, double rtol,
                          double atol, int maxit, int restart, int flags, tg_comm_t comm, int *iters,
                          double *resnorm, int *status);

/* generateM for a spline given by element-wise Bezier extraction operators (Rhino T-splines,
 * tIGAr/RhinoTSplines.py:37-137 + the row loop of tIGAr/common.py:1554-1571): FE row (e, n) holds
 * N_a = sum_b coef[a][b] * bern[e][n][b] for the functions a of element e (eoff[e] <= a < eoff[e+1], global index
 * nodes[a] + col_offset, ascending per element), entries with abs(v) <= eps dropped. */
int tg_extract_csr_bezier(int64_t nel, int nloc, int nbern, const double *bern, const int64_t *eoff,
                          const int32_t *nodes, const double *coef, int32_t col_offset, int64_t ncols, double eps,
                          tg_csr_t *out);

/* ---- direct solve: what solveLinearSystem runs when linearSolver is None (dolfin solve() = sparse LU [ext],
 * tIGAr/common.py:1255-1256).  Banded LU with partial pivoting in LAPACK's dgbtrf storage / pivoting scheme. */
/* half-bandwidths of a square matrix and the bytes its band storage (2*kl+ku+1 rows) would take */
int tg_lu_band_info(tg_csr_t k, int *kl, int *ku, int64_t *bytes);
/* x = K^-1 b (x may be b); info > 0: U(info-1,info-1) == 0 exactly, nothing was solved */
int tg_lu_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int *info);
/* The same system by a blocked banded Cholesky factorisation on the matrix cores (csrc/tg_chol.hip) when K is symmetric
 * (values compared with their transposes) and positive definite (every pivot positive): *done = 1 and x = K^-1 b (x may
 * be b); otherwise *done = 0 and x is untouched.  tg_lu_solve tries this itself first; the entry exists for systems
 * beyond the LU's limits (3-D patches: n kl^2 multiply-adds and n (kl + 1) doubles instead of 4 n kl^2 and 3 n kl), where
 * the caller's alternative is a Krylov method (tigar_amd.common._DefaultSolver). */
int tg_chol_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int *done);

/* ---- synthetic FE-side input (NOT on the timed path) --------- */
/* A = sum_t (x)_k F[t][k] with 1-D CSR factors sharing one pattern per direction
 * (e.g. Q_p Laplace stiffness = K1xM1xM1 + M1xK1xM1 + M1xM1xK1); rows [row0,row1). */
typedef struct {
  int64_t n;               /* 1-D size                                   */
  const int32_t *rowptr;   /* n+1                                        */
  const int32_t *col;      /* shared pattern                             */
  const double *val;       /* nterms * nnz1d values, term-major          */
} tg_kron_dir_t;
int tg_kron_sum_csr(int d, int nterms, const tg_kron_dir_t *dirs, int64_t row0,
                    int64_t row1, tg_csr_t *out);
/* The same planes for an FE matrix GIVEN AS A KRONECKER SUM of 1-D matrices (dirs[k]: the 1-D pattern of direction k
 * and nterms value sets, term-major -- the arguments of tg_kron_sum_csr): the matrix is never written, its entries are
 * formed inside the x pass exactly as tg_kron_sum_csr forms them, so the planes are bit for bit those of
 * tg_tensor_planes on the materialised matrix.  100: the 1-D patterns are not the element-coupling patterns of the
 * plan, or more than three terms (materialise and call tg_tensor_planes).  SURVEY 8(d): "If the build fuses A-generation
 * into PtAP (never materialising A) ...". */
int tg_tensor_planes_kron(tg_tensor_plan_t plan, int nterms, const tg_kron_dir_t *dirs, int z0, int z1,
                          tg_tensor_planes_t *out);

/* General Kronecker-product CSR builder on the device: out = sum_t (x)_k F[t][k] restricted to rows
 * [row0,row1), with rectangular 1-D factors (cdim[k] = number of columns of direction k; NULL =
 * square), optional |v| > eps filter (single term), a column offset and a total column count
 * (-1 = prod cdim).  Used for the directional extraction operators I(x)I(x)M_x ... of the
 * sum-factorised M^T A M and for their transposes. */
int tg_kron_csr_rect(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0,
                     int64_t row1, int filter, double eps, int64_t col_offset, int64_t ncols_total,
                     tg_csr_t *out);
/* generateM / its transpose for a tensor-product B-spline whose abs(v) > eps filter drops nothing but exact zeros
 * (the caller checks that on the 1-D tables): out = F[2] (x) F[1] (x) F[0] restricted to rows [row0,row1), values
 * (v0*v1)*v2 in the reference's order (tIGAr/BSplines.py:450-503), written by a pencil walk with closed-form row
 * starts (no count pass, no scan); cdim[k] = columns of direction k; columns shifted by col_offset. */
int tg_kron3_csr(int d, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                 int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

/* One 1-D sparse factor F (nout_k rows, CSR on the host, columns in [col_shift, col_shift+dims_in[k]))
 * applied along direction k of a tensor-indexed vector (direction 0 fastest):
 *     out[lo, I, hi] = sum_t F[I, c_t] in[lo, c_t - col_shift, hi].
 * Three such passes give M^T b (multTranspose, tIGAr/common.py:97-109) and M U (:1259) of a
 * Kronecker-structured extraction operator without forming M or M^T. */
int tg_tensor_apply_1d(int d, const int64_t *dims_in, int k, int64_t nout_k, const int32_t *rowptr,
                       const int32_t *col, const double *val, int64_t col_shift, tg_vec_t in, tg_vec_t out);

/* ---- FE-side operator assembly on mapped tensor-product patches ----
 * Stands in for dolfin.assemble(form) (tIGAr/common.py:1206-1220) with the spline's measures:
 * geometry F = cp[i]/cp[nsd] (tIGAr/common.py:917-921), metric g = DF^T DF, volume element
 * sqrt(det g) (tIGAr/calculusUtils.py:66-70), Cartesian gradient through pinv(DF)
 * (tIGAr/calculusUtils.py:56-64), Gauss-Legendre quadrature with nq points per direction
 * (quadrature_degree 2p <-> nq = p+1).  Scalar Q_p Lagrange space on the tensor node grid with
 * direction 0 fastest; nsd >= d (surfaces: Laplace-Beltrami). */
typedef struct {
  int d, p;                 /* parametric dimension, FE degree                        */
  const double *verts[3];   /* element vertices per direction (host), nverts[k] each  */
  int nverts[3];
  int nsd;                  /* physical dimension, d <= nsd <= 3                      */
  tg_vec_t cp[4];           /* nsd+1 homogeneous control functions on the FE nodes    */
  int nq;                   /* Gauss points per direction                             */
} tg_patch_t;
/* form: 0 = (u,v), 1 = (grad u, grad v), 4 = (lap u, lap v) element by element with lap = spline.div(spline.grad(.)) and
 * the second derivatives of the map (nsd == d; demos/biharmonic/biharmonic.py:100-103); result on the element-coupling
 * pattern */
int tg_assemble_mapped_matrix(const tg_patch_t *patch, int form, tg_csr_t *out);
/* L(v) = (f_h, v) with f_h the nodal interpolant of fnodal */
int tg_assemble_mapped_load(const tg_patch_t *patch, tg_vec_t fnodal, tg_vec_t out);
/* Row blocks of the same objects, as the z-slab pipeline of extractMatrix / extractVector consumes them (PETSc's row-block
 * MatPtAP, tIGAr/common.py:1194-1195; the assembled matrix of BASELINE cfg3 would hold 684 GB): rows [row0, row1) = whole
 * node planes of the LAST direction, global column indices, the element-coupling pattern with its certificate
 * (tg_tensor_planes then reads no column index).  patch->cp[c] (and fnodal) hold the FE nodes [cp_node0, cp_node0 + n) --
 * a window that must cover every element touching the rows -- so a rank never needs the control functions outside its
 * slab.  3-D patches with nq = p + 1 <= 4: sum-factorised element matrices, one wave per element
 * (csrc/tg_assemble.hip, k_asf3). */
int tg_assemble_mapped_matrix_rows(const tg_patch_t *patch, int form, int64_t row0, int64_t row1, int64_t cp_node0,
                                   tg_csr_t *out);
int tg_assemble_mapped_load_rows(const tg_patch_t *patch, tg_vec_t fnodal, int64_t row0, int64_t row1, int64_t cp_node0,
                                 tg_vec_t out);
/* Block (fi, fj) of linear elasticity a(u,v) = int lambda div u div v + 2 mu eps(u):eps(v) dx on the mapped patch, u, v in
 * the d-field space on the patch's node grid (nsd == d): lambda (d_fi phi_a, d_fj phi_b) + mu (d_fj phi_a, d_fi phi_b)
 * + delta mu (grad phi_a, grad phi_b) with the Cartesian derivatives of spline.grad / spline.div (tIGAr/common.py:1022-1040,
 * calculusUtils.py:255-276) -- what dolfin.assemble gives for inner(sigma(u), eps(v))*spline.dx restricted to test
 * component fi, trial component fj.  Rows and control-function window as tg_assemble_mapped_matrix_rows
 * (row0 = row1 = -1: the whole block); same pattern, same kernels (the coefficient tensor per point is not symmetric). */
int tg_assemble_mapped_elasticity_rows(const tg_patch_t *patch, int fi, int fj, double lambda, double mu, int64_t row0,
                                       int64_t row1, int64_t cp_node0, tg_csr_t *out);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) --------- */
int tg_comm_unique_id(char *id128);                          /* ncclGetUniqueId   */
int tg_comm_create(const char *id128, int rank, int world, tg_comm_t *out);
/* the same with a second RCCL communicator for the halo send/recv pairs (exchange stream), so that they and the
 * all-reduces (solver stream) never share one; id_halo may be NULL.  ncclCommInitRank is given TIGAR_RCCL_TIMEOUT_S
 * seconds (default 180) and reported as failed afterwards instead of blocking for ever. */
int tg_comm_create2(const char *id_reduce128, const char *id_halo128, int rank, int world, tg_comm_t *out);
/* Host-staged communicator: the same solver code, its two exchanges (halo pieces with the z-neighbours,
 * sums of a few doubles over all ranks) staged through pinned host memory and carried by the caller's
 * transport (MPI-style callbacks; 0 = ok).  For process groups RCCL cannot form (ranks sharing a GPU).
 * PETSc counterparts: VecScatter in MatMult, MPI_Allreduce in VecDot/VecNorm (tIGAr/common.py:1255-1258). */
typedef int (*tg_host_allreduce_fn)(void *ctx, double *inout, int n);
typedef int (*tg_host_sendrecv_fn)(void *ctx, int peer, const double *send, int64_t nsend, double *recv,
                                   int64_t nrecv);
int tg_comm_create_host(int rank, int world, tg_host_allreduce_fn allreduce, tg_host_sendrecv_fn sendrecv,
                        void *ctx, tg_comm_t *out);
/* IPC communicator: no host in the loop and no RCCL -- halo planes are pushed by a kernel into the neighbour's device
 * mailbox (hipIpcOpenMemHandle: the same GPU when ranks share one, an xGMI peer otherwise), flags and the slots of
 * the small all-reduce live in the shared file `shm_path` (created and sized to tg_comm_ipc_shm_bytes() zero bytes by
 * the launcher before any rank calls this; at most 16 ranks of one node), waits happen inside the kernels with a
 * wall-clock limit (TIGAR_IPC_TIMEOUT_S, default 60): an exchange is enqueue-only, a dead peer surfaces as an error
 * at the next host wait instead of hanging the GPU.  SURVEY 8(e): "direct xGMI peer copies". */
int tg_comm_ipc_shm_bytes(int64_t *bytes);
int tg_comm_create_ipc(const char *shm_path, int rank, int world, tg_comm_t *out);
int tg_comm_rank_device(tg_comm_t c, int rank, int *device);
/* ranks the communicator really spans (RCCL: ncclCommCount) and its kind (0 = RCCL, 1 = host-staged, 2 = IPC) */
int tg_comm_info(tg_comm_t c, int *rank, int *world, int *kind);
int tg_device_count(int *n);                                 /* visible GPUs      */
/* z-slab descriptor of the Krylov vectors: this rank owns global dofs [g0,g1); the SpMV
 * needs halo_lo dofs below g0 (owned by rank-1) and halo_hi above g1 (rank+1). */
int tg_comm_set_slab(tg_comm_t c, int64_t g0, int64_t g1, int64_t halo_lo, int64_t halo_hi,
                     int64_t nglobal);
int tg_comm_allreduce_sum(tg_comm_t c, double *host_inout, int n);
/* xext = [halo_lo | x_local | halo_hi] with the halos fetched from the z-neighbours */
int tg_comm_halo_extend(tg_comm_t c, tg_vec_t x_local, tg_vec_t xext);
int tg_comm_destroy(tg_comm_t c);
/* one small all-reduce and three halo exchanges with known values on a stream of their own, every host wait bounded by
 * timeout_s: 0 = works; 4 = an exchange did not complete (drop the communicator WITHOUT destroying it); else wrong
 * values / errors.  Collective; the launcher falls back to the next kind of communicator on failure. */
int tg_comm_selftest(tg_comm_t c, double timeout_s);

#ifdef __cplusplus
}
#endif
#endif
