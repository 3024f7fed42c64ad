/*
 * fsq.h - C ABI of libfsq_hip.so: the MI355X (gfx950) implementation of the reference's per-field
 * image hot path.  The reference (marcottelab/FluorosequencingImageAnalysis) is pure Python and has
 * no FFI; these entry points are what a binding for that path binds instead of the Python bodies:
 *
 *   fsq_find_peptides      <- pflib.find_peptides as a whole   pflib.py:284-520 (composes the four below; one call per batch)
 *   fsq_detect             <- pflib._psf_candidates            pflib.py:217-258
 *   fsq_fit_candidates     <- the candidate loop of pflib.find_peptides (pflib.py:441-477):
 *                             pflib._fit_2d_gaussian :180-214 -> gaussfitter.gaussfit
 *                             (agpy/gaussfitter.py:142-255) -> mpfit (agpy/mpfit/mpfit.py:600-1388),
 *                             then r_2 / rmse / illumina_s_n  (pflib.py:461-473, 261-281)
 *   fsq_consolidate        <- R^2 filter + consolidation + re-key   pflib.py:466, 479-519
 *   fsq_fit_images         <- gaussfitter.twodgaussian on the kept peaks  gaussfitter.py:253
 *   fsq_phase_correlate    <- phase_correlate.phase_correlate  phase_correlate.py:11-134
 *   fsq_fitq_*             <- the image loop around the fits    pflib.py:940-996, 1082-1099 (continuous batching)
 *   fsq_greedy_tracking    <- Experiment.greedy_particle_tracking (+ accumulate_offsets / discard_dropouts)  flexlibrary.py:567-1027
 *   fsq_centroid_tracking  <- Experiment.luminosity_centroid_particle_tracking  flexlibrary.py:1173-1317
 *   fsq_mexican_hat        <- Spot.mexican_hat_photometry_metric  flexlibrary.py:172-210
 *
 * Conventions: every function returns 0 on success or a negative FSQ_E* code; nothing throws or
 * aborts.  All pointers named d_* are DEVICE pointers (HBM); the caller owns every buffer.  `stream`
 * is a hipStream_t passed as void*; functions only enqueue work on it and never synchronise unless
 * stated.  Thread-safe for distinct streams/buffers.  INTEGRATION.md shows the Python-side binding.
 */
#ifndef FSQ_H
#define FSQ_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FSQ_OK         0
#define FSQ_EINVAL   (-1)   /* bad argument: the reference raises ValueError (pflib.py:238, 432) */
#define FSQ_ENOMEM   (-2)
#define FSQ_ERANGE   (-3)   /* output capacity too small; *needed is set */
#define FSQ_EHIP     (-4)   /* a HIP call failed (fsq_last_hip_error()) */
#define FSQ_EASSERT  (-5)   /* the reference's assert at pflib.py:518 would fire */
#define FSQ_ENOTIMPL (-6)   /* reference raises NotImplementedError (pflib.py:195) */
#define FSQ_EAGAIN   (-7)   /* FsqFitQueue: no room for the batch right now - advance the queue and submit again */
#define FSQ_EINTERNAL (-8)  /* FsqFitQueue: no fit is alive but a batch is incomplete - an engine bug, reported instead of waiting for ever */

#define FSQ_MODE_REF      0 /* reference-faithful fp64 LM (qrsolv/diag(R) aliasing of mpfit.py:1915) */
#define FSQ_MODE_TEXTBOOK 1 /* same solver with MINPACK's diagonal restore */
#define FSQ_MODE_TEXTBOOK_F32 2 /* OPT-IN approximation (BASELINE configs[4] "fp32 LM accumulate"): single-precision LM with an
                                 analytic Jacobian and 7x7 normal equations, same model / start / bounds (gaussfitter.py:100-136,
                                 pflib.py:199-212); results are NOT the reference's bit for bit - csrc/fsq_fit_f32.h, DESIGN.md 4.9.
                                 R^2 / rmse / s_n of the row are still computed in fp64 from the fitted parameters */
#define FSQ_PIXELS_F16_FLAG 0x1000 /* OR into mode (fsq_fit_candidates): d_img holds FSQ_PIXELS_F16 pixels */
#define FSQ_PIXELS_U32_FLAG 0x2000 /* OR into mode (fsq_fit_candidates): d_img holds FSQ_PIXELS_U32 pixels (uint32[n_fields][H][W]) */
#define FSQ_ENGINE_LANE 0x100 /* OR into mode: persistent kernel, one GPU lane per fit (A/B timing only) */
#define FSQ_ENGINE_QUAD 0x200 /* OR into mode: persistent kernel, a quad of lanes per fit (A/B timing only) */

/* One fitted candidate: pflib's PSF tuple (pflib.py:475) without the two 5x5 images. 128 bytes. */
typedef struct FsqRow {
    double h0, w0, H, A, sigma_h, sigma_w, theta;   /* tuple[0..6]  (h0 = p2 + h - 2.5, pflib.py:461) */
    double rmse, r2, s_n;                           /* tuple[9..11] */
    double p2, p3;                                  /* fitter-frame centre, to rebuild fit_img exactly */
    int32_t h, w, field;                            /* candidate pixel and field index */
    int32_t status, niter, nfev;                    /* mpfit exit status, iterations, model evaluations */
    int32_t key_h, key_w;                           /* dict key after consolidation (rounded centre) */
} FsqRow;

/*
 * Pixel storage formats of the image arguments.  The reference works on `image.astype(np.int64)` whatever dtype it is
 * handed (pflib.py:241, 443); the GPU path keeps images in HBM as 16-bit words:
 *   FSQ_PIXELS_U16  unsigned 16-bit integers (TIRF camera frames)
 *   FSQ_PIXELS_F16  IEEE binary16 holding the (pre-scaled, BASELINE.json configs[4]) intensities; a pixel's integer value
 *                   is the half truncated toward zero, exactly what astype(int64) gives for a float16 image
 *                   (negative / NaN -> 0, +inf -> 65535).  Same bytes per pixel, same fp64 solver behind the load.
 */
#define FSQ_PIXELS_U16 0
#define FSQ_PIXELS_F16 1
#define FSQ_PIXELS_U32 2   /* round 4: uint32 pixels (values < 2^31) for images beyond 16 bits - the reference computes on int64 whatever it
                            * is handed.  Taken by fsq_detect, fsq_fit_candidates (| FSQ_PIXELS_U32_FLAG), fsq_find_peptides (records of
                            * FSQ_PEAK_RECORD_BYTES_U32 bytes), a fit queue created for them (fsq_fitq_create with mode |
                            * FSQ_PIXELS_U32_FLAG: such a queue takes uint32 batches only, any other queue none) and fsq_fit_images /
                            * fsq_consolidate / fsq_kept_rows (format-independent); photometry and centroid tracking have _u32 twins
                            * (fsq_mexican_hat_u32, fsq_centroid_tracking_u32); only the single-precision mode stays 16-bit (FSQ_ENOTIMPL). */

#define FSQ_MAX_KSIZE 15                           /* largest correlation matrix / median window side (round 4: was 9) */
typedef struct FsqDetectParams {
    int32_t median_filter_size;                     /* pflib default 5; 1 .. FSQ_MAX_KSIZE */
    int32_t ksz;                                    /* correlation_matrix side, odd, <= FSQ_MAX_KSIZE */
    double c_std;                                   /* pflib default 2 */
    int64_t K[FSQ_MAX_KSIZE * FSQ_MAX_KSIZE];       /* correlation_matrix, row-major (ksz * ksz entries used) */
    int32_t pixel_format;                           /* FSQ_PIXELS_U16 / FSQ_PIXELS_F16 / FSQ_PIXELS_U32 of d_img */
    int32_t pixel_bits;                             /* FSQ_PIXELS_U32: significant bits of the largest pixel value (1 .. 31; 0 = 31): the
                                                     * exactness domain of the integer response is checked against it */
} FsqDetectParams;

const char* fsq_version(void);
const char* fsq_last_hip_error(void);
int fsq_device_count(void);

/* Workspace bytes fsq_detect needs for n_fields fields of H x W. */
int64_t fsq_detect_workspace_bytes(int n_fields, int H, int W);

/*
 * Candidate detection for a batch of fields (raster order inside each field, fields in order).
 *   d_img     uint16[n_fields][H][W] (or binary16, prm->pixel_format)
 *   d_cand    int32[cap][3]  (field, h, w)            out
 *   d_counts  int32[n_fields + 1]                      out: per-field candidate counts, [n_fields] = total
 *             (total = -1: the response image of some field sums to >= 2^53, where numpy.mean of the reference
 *             stops being the exact integer mean this kernel computes - treat as "not implemented")
 *   d_offsets int32[n_fields + 1]                      out: exclusive prefix of d_counts (field f's
 *                                                           candidates are d_cand[offsets[f] .. +counts[f]))
 *   d_thr     double[n_fields] (may be NULL)           out: mean + c_std * std of the response image
 * If the total exceeds cap only the first cap candidates are written; d_counts/d_offsets are still
 * complete, so the caller can re-run with a larger buffer.  Enqueue only.
 */
int fsq_detect(const uint16_t* d_img, int n_fields, int H, int W, const FsqDetectParams* prm,
               int32_t* d_cand, int64_t cap, int32_t* d_counts, int32_t* d_offsets, double* d_thr,
               void* d_workspace, int64_t workspace_bytes, void* stream);

/* Workspace bytes the fit entry points need for n candidates. */
int64_t fsq_fit_workspace_bytes(int64_t n);

/* LM-fit n candidates (any mix of fields); d_rows[n] out.
 * mode: FSQ_MODE_REF / FSQ_MODE_TEXTBOOK (/ FSQ_MODE_TEXTBOOK_F32: one persistent launch, no rounds).  The default engine advances all candidates in rounds
 * (Jacobian round / step round, see csrc/fsq_fit_rounds.hip): the call drives those rounds from the calling
 * host thread and synchronises the stream every few rounds to read the queue sizes, so it returns when the
 * last round has run (large batches finish their last, nearly empty rounds on an internal highest-priority
 * stream); only the final row-writing kernel is merely enqueued on `stream`.  Thread-safe: concurrent calls
 * with different workspaces / streams are independent (engine.LanePipeline).
 * | FSQ_ENGINE_LANE or | FSQ_ENGINE_QUAD select the two single-launch persistent engines instead (identical
 * results; kept for A/B timing). */
int fsq_fit_candidates(const uint16_t* d_img, int n_fields, int H, int W, const int32_t* d_cand, int64_t n,
                       int mode, FsqRow* d_rows, void* d_workspace, int64_t workspace_bytes, void* stream);

/*
 * FsqFitQueue - the LM-fit engine kept alive ACROSS batches (continuous batching).
 *
 * Replaces the same reference code as fsq_fit_candidates (the candidate loop of pflib.find_peptides, pflib.py:441-477,
 * down to mpfit); its reference-side counterpart as a whole is the image loop around that: pflib.image_batch /
 * parallel_image_batch (pflib.py:940-996, 1082-1099), which works images one after the other (or one per process).
 * One LM solve needs 1..200 sequential iterations, so a stand-alone batch ends in a long tail of rounds in which a
 * few per cent of its fits are alive and the chip idles.  A queue keeps the solver state of every
 * fit in flight in one pool: batches are submitted while earlier ones are still finishing, every round advances all of
 * them in the same full launches, and a batch's rows are written as soon as its last fit has terminated.  Results are
 * bit-identical to fsq_fit_candidates (fits are independent; only the order of execution changes).
 *
 *   pool_slots  candidates in flight at most (every batch owns n slots from submit until its rows are written;
 *               slots are handed out in ring order, so leave room for about two batches more than are in flight)
 *   queue_cap   fits ALIVE at most (sum over the batches in flight of their unfinished fits)
 * The queue lives on one stream: submit / advance only enqueue on it, except that advance synchronises it every
 * few rounds to read the queue sizes.  Not thread-safe: one host thread drives a queue.  At most FSQ_MAX_TICKETS
 * batches in flight.
 */
#define FSQ_MAX_TICKETS 32
typedef struct FsqFitQueue FsqFitQueue;
int64_t fsq_fitq_workspace_bytes(int64_t pool_slots, int64_t queue_cap);
int fsq_fitq_create(FsqFitQueue** q, void* d_workspace, int64_t workspace_bytes, int64_t pool_slots, int64_t queue_cap,
                    int mode /* FSQ_MODE_REF / FSQ_MODE_TEXTBOOK / FSQ_MODE_TEXTBOOK_F32 */, void* stream);
/* Add a batch (same arguments as fsq_fit_candidates; d_img / d_cand / d_rows must stay valid until the batch has been
 * taken).  Work enqueued on the queue's stream so far must not still be writing d_cand (order it with an event or
 * synchronise).  *ticket names the batch.  FSQ_EAGAIN: no free slots / tickets now. */
int fsq_fitq_submit(FsqFitQueue* q, const void* d_img, int pixel_format, int n_fields, int H, int W, const int32_t* d_cand,
                    int64_t n, FsqRow* d_rows, int* ticket);
/* Run rounds until at least one batch has finished, or nothing is alive, or fewer than alive_below fits are alive, or
 * max_rounds (> 0) rounds have run.  *alive = fits still alive, *finished = batches that finished during the call. */
int fsq_fitq_advance(FsqFitQueue* q, int64_t max_rounds, int64_t alive_below, int64_t* alive, int* finished);
/* 1: the batch has finished - consumer_stream is made to wait for its rows, the ticket is free again;
 * 0: still in flight; < 0: error. */
int fsq_fitq_take(FsqFitQueue* q, int ticket, void* consumer_stream);
int64_t fsq_fitq_alive(const FsqFitQueue* q);
int64_t fsq_fitq_rounds(const FsqFitQueue* q);
int fsq_fitq_destroy(FsqFitQueue* q);      /* synchronises the queue's stream */

/* Fit n stand-alone ROIs uint16[n][25] (pflib._fit_2d_gaussian surface); h = w = 2, field = 0. */
int fsq_fit_rois(const uint16_t* d_rois, int64_t n, int mode, FsqRow* d_rows, void* d_workspace,
                 int64_t workspace_bytes, void* stream);

int64_t fsq_consolidate_workspace_bytes(int n_fields, int H, int W);
/*
 * R^2 filter + consolidation + re-key, per field, sequential reference semantics.
 *   d_rows     FsqRow[n]: all fitted candidates, grouped by field in raster order (as fsq_detect emits)
 *   d_counts, d_offsets   int32[n_fields+1] as fsq_detect emits them
 *   d_keep     int32[n]  out: indices into d_rows of kept peaks; field f's kept indices are
 *              d_keep[offsets[f] .. offsets[f] + nkeep[f]), in the reference's dict order
 *   d_nkeep    int32[n_fields+1] out: kept per field, [n_fields] = total; -1 for a field whose
 *              re-key assertion (pflib.py:518) fired
 * key_h/key_w of the kept rows are filled in.
 */
int fsq_consolidate(FsqRow* d_rows, const int32_t* d_counts, const int32_t* d_offsets, int n_fields, int H, int W,
                    double r2_threshold,
                    int radius, int py2_round, int32_t* d_keep, int32_t* d_nkeep, void* d_workspace,
                    int64_t workspace_bytes, void* stream);

/*
 * The kept peaks of all fields as ONE contiguous table in field order (what pflib.find_peptides returns per image,
 * pflib.py:520, concatenated) - the unit that is gathered across GPUs and copied to the host.
 *   d_rows / d_keep / d_offsets / d_nkeep   as fsq_consolidate leaves them
 *   d_out          FsqRow[cap] out; rows beyond cap are dropped (d_nkeep[n_fields] is the total kept)
 *   d_out_offsets  int32[n_fields + 1] out: field f's peaks are d_out[out_offsets[f] .. out_offsets[f+1])
 *                  (a field whose re-key assertion fired, nkeep = -1, contributes none)
 * Enqueue only.
 */
int fsq_kept_rows(const FsqRow* d_rows, const int32_t* d_keep, const int32_t* d_offsets, const int32_t* d_nkeep,
                  int n_fields, FsqRow* d_out, int64_t cap, int32_t* d_out_offsets, void* stream);

/*
 * THE WHOLE PATH IN ONE CALL: pflib.find_peptides (pflib.py:284-520) for every field of a batch already in HBM - candidate
 * detection (:217-258), the LM fit of every candidate (:441-475), R^2 filter + consolidation + re-keying (:466, 477-519) -
 * returning what the reference's dict holds per kept peak as a flat record table.  This is the entry point a binding of the
 * reference's find_peptides(image) -> dict would use; it composes fsq_detect, fsq_fit_candidates,
 * fsq_consolidate, fsq_kept_rows and fsq_fit_images on the caller's stream and workspace.
 *   d_img              uint16 / binary16 [n_fields][H][W] (prm->pixel_format)
 *   mode               FSQ_MODE_REF / FSQ_MODE_TEXTBOOK / FSQ_MODE_TEXTBOOK_F32
 *   cand_cap           candidates the workspace is sized for (all fields together)
 *   d_records          uint8[record_cap][FSQ_PEAK_RECORD_BYTES] (FSQ_PEAK_RECORD_BYTES_U32 for uint32 pixels) out, fields in order, a field's peaks in the reference's dict
 *                      order: bytes 0..127 the FsqRow, 128..327 fit_img double[25] (gaussfitter.py:253), 328..377 the 25
 *                      16-bit pixel words of sub_img as they sit in d_img (pflib.py:443)
 *   d_record_offsets   int32[n_fields + 1] out: field f's records are [offsets[f], offsets[f + 1])
 *   d_nkeep            int32[n_fields + 1] out: kept peaks per field, -1 where the reference's assert (pflib.py:518) would
 *                      fire (such a field has no records); [n_fields] = total
 *   n_candidates, n_records   host out (may be NULL)
 * Returns FSQ_ERANGE when cand_cap or record_cap is too small - *n_candidates / *n_records then hold what is needed and the
 * call can be repeated with a larger workspace.  Synchronises the stream twice (the candidate total and the kept total size
 * the following launches); the records themselves are complete when the work enqueued on `stream` is.
 */
#define FSQ_PEAK_RECORD_BYTES 378
#define FSQ_PEAK_RECORD_BYTES_U32 428   /* prm->pixel_format == FSQ_PIXELS_U32: the same record with sub_img as 25 uint32 words */
int64_t fsq_find_peptides_workspace_bytes(int n_fields, int H, int W, int64_t cand_cap, int64_t record_cap);
int fsq_find_peptides(const void* d_img, int n_fields, int H, int W, const FsqDetectParams* prm, double r2_threshold,
                      int radius, int py2_round, int mode, int64_t cand_cap, void* d_records, int64_t record_cap,
                      int32_t* d_record_offsets, int32_t* d_nkeep, int64_t* n_candidates, int64_t* n_records,
                      void* d_workspace, int64_t workspace_bytes, void* stream);

/* fit_img (double[n][25]) of rows selected by d_idx[n] (NULL = all first n rows). */
int fsq_fit_images(const FsqRow* d_rows, const int32_t* d_idx, int64_t n, double* d_fit_img, void* stream);

/*
 * Registration of n_pairs image pairs: out4 = (row_shift, col_shift, error, diffphase) per pair, double[n_pairs][4]
 * (phase_correlate.phase_correlate, phase_correlate.py:11-134; the reference converts whatever it is handed to float64,
 * :63-64).
 *   d_ref, d_reg  [n_pairs][H][W] of `dtype`: FSQ_DTYPE_F64 (double) or FSQ_DTYPE_U16 (camera frames as they sit in HBM)
 *   d_workspace   fsq_phase_correlate_workspace_bytes(...) bytes for the same arguments (spectra, FFT work areas, peaks)
 * Enqueue only: FFTs (rocFFT real-to-complex / complex-to-real, fp64), the peak search, the upsampled matrix-multiply DFT
 * and the error / phase arithmetic all run on `stream`; nothing is allocated and the host never waits.
 * FFT plans are cached per (shape, batch, device, stream); calls are serialised on an internal lock while they enqueue.
 */
#define FSQ_DTYPE_F64 0
#define FSQ_DTYPE_U16 1
int64_t fsq_phase_correlate_workspace_bytes(int n_pairs, int H, int W, int upsample_factor, int dtype, void* stream);
int fsq_phase_correlate(const void* d_ref, const void* d_reg, int dtype, int n_pairs, int H, int W, int upsample_factor,
                        double* d_out4, void* d_workspace, int64_t workspace_bytes, void* stream);

/*
 * Spot photometry on the peak table: Spot.mexican_hat_photometry_metric, flexlibrary.py:172-210.
 *   d_img   uint16[n_fields][H][W]; d_fhw int32[n][3] = (field, h, w) integer spot centres (the caller guarantees
 *           0 <= field < n_fields; h, w may lie anywhere - the window is clipped like Spot.image_slice,
 *           flexlibrary.py:140-146, and an empty brim gives nan like numpy.median([]))
 *   d_out   double[n]: sum(crown) - len(crown) * median(brim); any radius (windows beyond 31 x 31 are re-read per median step)
 * Enqueues on the stream, does not synchronise.
 */
int fsq_mexican_hat(const uint16_t* d_img, int n_fields, int H, int W, const int32_t* d_fhw, int64_t n,
                    int brim_size, int radius, double* d_out, void* stream);
/* the same on FSQ_PIXELS_U32 frames (uint32[n_fields][H][W], values < 2^31) */
int fsq_mexican_hat_u32(const uint32_t* d_img, int n_fields, int H, int W, const int32_t* d_fhw, int64_t n,
                        int brim_size, int radius, double* d_out, void* stream);

/*
 * Greedy particle tracking of the peak tables across the frames of a field:
 * Experiment.greedy_particle_tracking with accumulate_offsets and discard_dropouts, flexlibrary.py:567-1027.
 * One call tracks n_fields independent fields of n_frames frames each.
 *   d_hw           int32[total][2]   Spot.h, Spot.w (the dict keys pflib.find_peptides returns = FsqRow.key_h / key_w),
 *                                    field after field, frame after frame inside a field
 *   d_field_start  int32[n_fields+1] first spot of every field in d_hw ([n_fields] = total)
 *   d_counts       int32[n_fields][n_frames]  spots per frame
 *   d_offsets      double[n_fields][n_frames][2]  (d_h, d_w) of every frame relative to the frame before it
 *                                    (SequenceExperiment.offsets_from_frames, flexlibrary.py:1717-1741); [0] must be (0, 0)
 *   candidate_radius, spot_radius    as the reference's arguments (defaults 2 and 0)
 * Outputs, spot numbers counted from the field's first spot, -1 = none:
 *   d_prev, d_next int32[total]      ancestor / descendant link of every spot
 *   d_kept         uint8[total]      0 = discarded because it drifts out of some frame (discard_dropouts)
 *   d_traces       int32[total][n_frames]  field f's traces are rows field_start[f] .. +n_traces[f], in the reference's
 *                                    order (heads by frame, then by bin in raster order); one spot number or -1 per frame
 *   d_n_traces, d_n_discarded, d_status  int32[n_fields]; status: 0, FSQ_EINVAL (offsets[0] != (0,0): the reference's
 *                                    ValueError), FSQ_EASSERT (two spots of a frame in one bin: its AssertionError,
 *                                    flexlibrary.py:851), FSQ_ERANGE (more than pair_cap candidate pairs in one frame)
 *   pair_cap       capacity of the per-frame candidate-pair list (a few times the spots per frame)
 * Any number of frames and spots (time series of more than 64 frames keep their frame tables in the workspace; fields of more
 * than 32 768 spots read the pairing facts off the links instead of LDS bitmaps).  Enqueue only.
 */
int64_t fsq_track_workspace_bytes(int n_fields, int n_frames, int H, int W, int64_t pair_cap);
int fsq_greedy_tracking(const int32_t* d_hw, const int32_t* d_field_start, const int32_t* d_counts, const double* d_offsets,
                        int n_fields, int n_frames, int H, int W, int candidate_radius, double spot_radius,
                        int32_t* d_prev, int32_t* d_next, uint8_t* d_kept, int32_t* d_traces, int32_t* d_n_traces,
                        int32_t* d_n_discarded, int32_t* d_status, int64_t pair_cap, void* d_workspace,
                        int64_t workspace_bytes, void* stream);
/*
 * Luminosity-centroid tracking: Experiment.luminosity_centroid_particle_tracking with
 * next_frame_spot_by_luminosity_centroid, flexlibrary.py:1173-1317, for Spots of size 5.
 *   d_frames      uint16[n_fields][n_frames][H][W]
 *   d_init_hw     int32[n][2]   the initial Spots (h, w) in frame 0 of their field; d_spot_field int32[n] their field
 *   d_offsets     int64[n_fields][n_frames][2] or NULL: offsets[f] is taken off the last sighting's coordinates when the
 *                 spot is looked for in frame f (whole pixels: the reference slices the image with them)
 *   d_out_hw      int32[n][n_frames][2] out: the spot's (h, w) in every frame, (-1, -1) where the reference has None
 *   d_present     uint8[n][n_frames] out
 *   d_n_errors    int32[1] out: spots whose search window summed to zero (the reference raises ValueError there:
 *                 int(round(nan))); their later frames are reported as absent
 * Enqueue only.
 */
int fsq_centroid_tracking(const uint16_t* d_frames, int n_fields, int n_frames, int H, int W, const int32_t* d_init_hw,
                          const int32_t* d_spot_field, int64_t n, int search_radius, double s_n_cutoff,
                          const int64_t* d_offsets, int32_t* d_out_hw, uint8_t* d_present, int32_t* d_n_errors, void* stream);
/* the same on FSQ_PIXELS_U32 frames (uint32[n_fields][n_frames][H][W], values < 2^31; search_radius <= 512) */
int fsq_centroid_tracking_u32(const uint32_t* d_frames, int n_fields, int n_frames, int H, int W, const int32_t* d_init_hw,
                              const int32_t* d_spot_field, int64_t n, int search_radius, double s_n_cutoff,
                              const int64_t* d_offsets, int32_t* d_out_hw, uint8_t* d_present, int32_t* d_n_errors, void* stream);
/* fsq_selftest_dnrm2: the tracking kernel's pair distance (OpenBLAS dnrm2 in x87 extended precision, restated in
 * integer arithmetic, csrc/fsq_x87.h) for caller-supplied displacement vectors; d_out[i] = dnrm2((d_dh[i], d_dw[i])). */
int fsq_selftest_dnrm2(const double* d_dh, const double* d_dw, int64_t n, double* d_out, void* stream);

/*
 * Self-test hooks of the LM fit (no reference counterpart).
 * fsq_selftest_division: the fit kernel divides by shared divisors through a hoisted reciprocal that is
 *   bit-identical to the compiler's fp64 division inside a guarded operand range (fsq_devmath.h); this counts
 *   the operand pairs (d_num[i], d_den[i]) for which the two differ.  Synchronises the stream.
 * fsq_fit_last_slow_count: number of fits of the last fsq_fit_* call that left the guarded range and were
 *   redone by the plain-division build of the kernel (env FSQ_DEBUG_FORCE_SLOW=k forces idx % k == 0 there).
 */
int fsq_selftest_division(const double* d_num, const double* d_den, int64_t n, int64_t* mismatches, void* stream);
/* fsq_selftest_rotation: qrsolv's 0.5 / sqrt(.25 + .25 t^2) (|t| <= 1) is evaluated by the cores of the compiler's
 * sqrt and division expansions (fsq_devmath.h); counts the d_t[i] for which that differs from the plain expression. */
int fsq_selftest_rotation(const double* d_t, int64_t n, int64_t* mismatches, void* stream);
/* fsq_selftest_exp: the fit kernels' branch-free exp (fsq_exp_bf) against the branching restatement of glibc's exp
 * (fsq_exp) - bit equality for |x| < 512; everywhere else (NaN included) the range flag must be raised instead. */
int fsq_selftest_exp(const double* d_x, int64_t n, int64_t* mismatches, void* stream);
/* fsq_selftest_square: qrfac's norm down-dating needs libm's pow(t, 2.0) (mpfit.py:1816, a NumPy scalar ** 2); the
 * Jacobian kernel takes t * t wherever that provably is the same number (fsq_square_is_pow2: t^2 further than pow's
 * own error bound from a rounding boundary).  Counts the d_t[i] for which the predicate holds and pow(t, 2.0) != t * t
 * (must be 0), and those for which it does not hold (*undecided; about 3 % of random arguments). */
int fsq_selftest_square(const double* d_t, int64_t n, int64_t* mismatches, int64_t* undecided, void* stream);
int64_t fsq_fit_last_slow_count(void);
/* 1 when the library was built with the two single-launch A/B engines (make AB target, -DFSQ_BUILD_AB): only then do
 * FSQ_ENGINE_LANE / FSQ_ENGINE_QUAD select them; the shipped library returns FSQ_ENOTIMPL for those flags. */
int fsq_has_ab_engines(void);

#ifdef __cplusplus
}
#endif
#endif
