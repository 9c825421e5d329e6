/* libimgan_hip.so -- C ABI of the MI355X (gfx950) SLIC / RAG / MPEG-7 hot path.
 *
 * The reference (filip141/Image-analysis) has no FFI: its boundary is the set of
 * Python classes in scripts/{slic,rag_segmentation,mpeg7Descriptors}.py.  Each
 * entry point below cites the reference code it replaces; the Python classes of
 * the same names in image-analysis_amd/ bind these symbols through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (IMGAN_E_*); the message
 *    is available from imgan_last_error() (thread local).
 *  - no ownership transfer: the caller allocates every buffer.  Pointers are
 *    DEVICE pointers on the current HIP device unless the name says `host`.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    nothing synchronises unless stated.  Calls are thread-safe per stream as
 *    long as they do not share a workspace.
 *  - images are u8, interleaved, batch first: [B][H][W][3] (cv2 BGR order);
 *    label maps are int32 [B][H][W] with -1 = unassigned (the reference keeps
 *    float64 label maps; the Python layer converts at the boundary).
 */
#ifndef IMGAN_H
#define IMGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMGAN_OK 0
#define IMGAN_E_INVALID (-1)
#define IMGAN_E_HIP (-2)
#define IMGAN_E_RANGE (-3)
#define IMGAN_E_NOMEM (-4)

typedef void* imgan_stream_t;

const char* imgan_last_error(void);
int imgan_version(void);

/* ---- colour conversion ---------------------------------------------------
 * Replaces cv2.cvtColor(img, COLOR_BGR2Lab) at slic.py:13,
 * rag_segmentation.py:12, mpeg7Descriptors.py:22 and COLOR_BGR2GRAY at
 * mpeg7Descriptors.py:23 (8-bit fixed-point OpenCV algorithm). */
int imgan_lab_tables_host(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9);
int imgan_bgr2lab_u8(const uint8_t* bgr, uint8_t* lab, int64_t npix, imgan_stream_t stream);
int imgan_bgr2gray_u8(const uint8_t* bgr, uint8_t* gray, int64_t npix, imgan_stream_t stream);

/* Replaces cv2.medianBlur(image, 7) at rag_segmentation.py:11 (3 channels,
 * replicated border). */
int imgan_median7_u8c3(const uint8_t* src, uint8_t* dst, int B, int H, int W,
                       imgan_stream_t stream);
/* cv2.medianBlur(image, ksize) for the other kernel sizes RAGSegmentation's median_blur argument
 * may carry (rag_segmentation.py:10-11): any odd ksize from 1 to 255 (3, 5 and 7 take the
 * bit-sliced kernels, the rest a plain per-pixel search; IMGAN_E_INVALID for an even size). */
int imgan_median_u8c3(const uint8_t* src, uint8_t* dst, int B, int H, int W, int ksize,
                      imgan_stream_t stream);

/* Replaces cv2.resize(img, (Wd, Hd)) (default INTER_LINEAR, 8 bit) at optigen_utils.py:23 and
 * basegen.py:30: OpenCV's fixed-point bilinear filter with 11-bit weights; an exact 2 x 2
 * reduction goes to the INTER_AREA rule as in OpenCV.  src [B][Hs][Ws][3], dst [B][Hd][Wd][3]. */
int imgan_resize_linear_u8c3(const uint8_t* src, uint8_t* dst, int B, int Hs, int Ws, int Hd, int Wd,
                             imgan_stream_t stream);

/* ---- SLIC ------------------------------------------------------------------
 * imgan_slic_grid: slic.py:29,33 (step) and :42-43 (seed grid); host only.
 * step_in <= 0 selects the default int(sqrt(H*W / K)). */
int imgan_slic_grid_host(int H, int W, int K, int step_in, int* step, int* nx, int* ny);

/* Bytes of device workspace the SLIC calls need for B images of H x W with n centres. */
size_t imgan_slic_workspace_bytes(int B, int H, int W, int S, int n);

/* slic.py:35-54 + :57-76 (init_clusters / find_local_minimum) and :81 (labels = -1).
 * centres: int32 [B][n][5] = L,a,b,x,y, n = nx*ny, index = ix*ny + iy. */
int imgan_slic_init(const uint8_t* lab, int B, int H, int W, int S, int nx, int ny,
                    int32_t* centres, int32_t* labels, void* workspace, imgan_stream_t stream);

/* The seeding half of imgan_slic_init alone (slic.py:35-76): centres only, no label map.  For
 * callers that run the first round with imgan_slic_assign_accumulate_first and so never need
 * the map filled with -1 (slic.py:81) in a pass of its own. */
int imgan_slic_seed_centres(const uint8_t* lab, int B, int H, int W, int S, int nx, int ny,
                            int32_t* centres, imgan_stream_t stream);

/* Candidate lookup for the next assignment: re-bins the centres into S x S cells and writes,
 * for every 64-pixel-wide tile, the sorted list of centres whose window (slic.py:94-100) can
 * reach it, with the per-centre terms of the distance of slic.py:101-109 (hence the colour
 * weight).  The reference has no counterpart: it loops over centres instead (slic.py:88). */
int imgan_slic_bin_centres(const uint8_t* lab, const int32_t* centres, int B, int H, int W, int S,
                           int n, double colour_weight, void* workspace, imgan_stream_t stream);

/* slic.py:85-116 (assignment over the 2S x 2S windows) fused with the sums of
 * :118-125.  colour_weight: slic.py:109.  Labels are updated in place. */
int imgan_slic_assign_accumulate(const uint8_t* lab, int32_t* labels, int B, int H, int W, int S,
                                 int n, double colour_weight, void* workspace,
                                 imgan_stream_t stream);

/* The same, and the bounding box of every label of the updated map on the side (slic.py has no
 * counterpart; the callers take them per segment with cv2.boundingRect, spatialextract.py:47,
 * or np.where, mpeg7Descriptors.py:28-33): bbox_out int32 [B][n][4] = xmin, ymin, xmax, ymax,
 * a label without pixels gets {INT_MAX, INT_MAX, -1, -1} as in imgan_label_bbox. */
int imgan_slic_assign_accumulate_bbox(const uint8_t* lab, int32_t* labels, int B, int H, int W,
                                      int S, int n, double colour_weight, void* workspace,
                                      int32_t* bbox_out, imgan_stream_t stream);

/* The first round after imgan_slic_seed_centres: as imgan_slic_assign_accumulate (with
 * bbox_out, which may be NULL, as imgan_slic_assign_accumulate_bbox) on a label map that counts
 * as all -1 (slic.py:81) whatever it holds: the map is written in full, not read, which saves
 * the pass that would fill it. */
int imgan_slic_assign_accumulate_first(const uint8_t* lab, int32_t* labels, int B, int H, int W,
                                       int S, int n, double colour_weight, void* workspace,
                                       int32_t* bbox_out, imgan_stream_t stream);

/* slic.py:124-126: centre = floor(sum / count), empty cluster -> 0. */
int imgan_slic_finalise_centres(int32_t* centres, int B, int H, int W, int S, int n,
                                void* workspace, imgan_stream_t stream);

/* counts int32 [B][n]: the pixels that carry each label after the last
 * imgan_slic_assign_accumulate* call (the divisor of slic.py:125), valid until the next
 * imgan_slic_bin_centres clears the sums.  What imgan_dcd_segments_listed wants to know. */
int imgan_slic_label_counts(int B, int H, int W, int S, int n, const void* workspace,
                            int32_t* counts, imgan_stream_t stream);

/* sums uint64 [B][K][4] = sum L, sum a, sum b, count per label after the last
 * imgan_slic_assign_accumulate* call, in imgan_rag_accumulate's layout (K >= 1; rows n..K-1 zero):
 * rag_segmentation.py:38-48 on the image SLIC ran on (run_slic :21-25 does run on image_lab), so
 * the RAG step of that image needs no second pass over its pixels.  Same validity as the counts. */
int imgan_slic_label_sums(int B, int H, int W, int S, int n, const void* workspace, int K,
                          uint64_t* sums, imgan_stream_t stream);

/* SLICSuperPixel.run(), slic.py:78-127: init + (num_iter-1) x (bin, assign, finalise). */
int imgan_slic_run(const uint8_t* lab, int B, int H, int W, int S, int nx, int ny,
                   double colour_weight, int num_iter, int32_t* centres, int32_t* labels,
                   void* workspace, imgan_stream_t stream);

/* ---- RAG -------------------------------------------------------------------
 * rag_segmentation.py:38-48 (per-label colour sums) and :50-74 (8-connected
 * adjacency) in one pass.  sums: uint64 [B][K][4] = sum c0,c1,c2,count.
 * edges: per image an open-addressed hash set of uint64 keys
 * ((min+1)<<32 | (max+1)), 0 = empty, capacity `edge_cap` (power of two);
 * must be zero-filled by the caller (as must sums).  overflow: int32 [B], zero-filled by
 * the caller, set to 1 if an image's edge set did not fit.  edges may be NULL (sums only);
 * img and sums may both be NULL (adjacency only: the image is not read -- for callers that have
 * the sums already, imgan_slic_label_sums).  Labels >= K are ignored for sums. */
int imgan_rag_accumulate(const uint8_t* img, const int32_t* labels, int B, int H, int W, int K,
                         uint64_t* sums, uint64_t* edges, int64_t edge_cap, int32_t* overflow,
                         imgan_stream_t stream);

/* rag_segmentation.py:77-184 on the host: edge weights, stable sort, Kruskal,
 * leave-one-out mean + c*sigma cut, Kruskal again, 1-based ids in surviving-set order.
 * All pointers are HOST pointers.  Two modes:
 *   means != NULL (int32 [K][3] floor means), weights == NULL: weights are the euclidean
 *     distances of the means (find_edges, :77-87); edges_uv in any order, duplicates allowed,
 *     taken in canonical order (u < v, lexicographic).
 *   weights != NULL (double [E]), means == NULL: weights as given; edges_uv must already be
 *     in the caller's discovery order (__convert2arrays, :90-100) without duplicates.
 * Vertices are 0..K-1 in list order.  lut_out: int32 [K] new id per vertex; mst_out: int32
 * [<=K][2] kept edges, mst_w_out their weights; members_out: int32 [K] the vertices set by
 * set in the reference's surviving-list order, set_offsets_out: int32 [n_sets+1].
 * mst_out, mst_w_out, members_out, set_offsets_out may be NULL. */
int imgan_rag_merge_host(const int32_t* means, const double* weights, int K,
                         const int32_t* edges_uv, int64_t E, double c_factor, int32_t* lut_out,
                         int32_t* mst_out, double* mst_w_out, int32_t* n_mst_out,
                         int32_t* members_out, int32_t* set_offsets_out, int32_t* n_sets_out);

/* The same merge (means mode, relabel table only) for every image of a batch, on up to
 * `threads` host threads (rag_segmentation.py:77-184 per image; the caller's loop over images,
 * basegen.py:96-99).  HOST pointers: means int32 [B][K][3]; edges_uv int32 [edge_offsets[B]][2],
 * image b owns edges edge_offsets[b] .. edge_offsets[b+1]-1; lut_out int32 [B][K]. */
int imgan_rag_merge_host_batch(const int32_t* means, int B, int K, const int32_t* edges_uv,
                               const int64_t* edge_offsets, double c_factor, int32_t* lut_out,
                               int threads);

/* The same merge (means mode, relabel table only) on the device, one workgroup per image, for
 * batches that stay on the GPU: reads the sums and the edge hash sets exactly as
 * imgan_rag_accumulate left them (DEVICE pointers, same B, K, edge_cap, overflow) and writes
 * lut_out int32 [B][K].  K <= 4096 (IMGAN_E_RANGE beyond: use the host form).  status: int32 [B]
 * (device), 0 = done; nonzero = this image's table was not written and the caller finishes it
 * with imgan_rag_merge_host (1: more than 16,384 edges, 2: an edge endpoint >= K, 3: the edge set
 * had overflowed).  The cut (rag_segmentation.py:163-170) is evaluated from the sum and the sum
 * of squares of the tree's weights in double-double arithmetic; a weight within ~1e-15 of its
 * threshold may fall on the other side than numpy's two-pass float64 mean / std puts it. */
int imgan_rag_merge_device(const uint64_t* sums, const uint64_t* edges, int64_t edge_cap,
                           const int32_t* overflow, int B, int K, double c_factor,
                           int32_t* lut_out, int32_t* status, imgan_stream_t stream);

/* rag_segmentation.py:175-182: out[p] = lut[labels[p]] (labels[p] < 0 or >= K -> 0).
 * lut: int32 [B][K] device. */
int imgan_relabel_lut(const int32_t* labels, const int32_t* lut, int32_t* out, int B,
                      int64_t npix_per_image, int K, imgan_stream_t stream);

/* rag_segmentation.py:34,46: paint the per-label mean colour back into the image. */
int imgan_paint_means(uint8_t* img, const int32_t* labels, const uint8_t* means_u8, int B,
                      int64_t npix_per_image, int K, imgan_stream_t stream);

/* ---- connected components ---------------------------------------------------
 * Label-propagation union-find over equal-valued 8-connected pixels
 * (mpeg7Descriptors.py:40-74,118-128 flood fill).  values: int32 [B][H][W],
 * values < 0 are background.  roots: int32 [B][H][W] = smallest raster index of
 * the component (image relative), -1 for background. */
int imgan_cc_label(const int32_t* values, int32_t* roots, int B, int H, int W,
                   imgan_stream_t stream);

/* Connectivity enforcement of a superpixel label map.  No counterpart in scripts/slic.py (it ends
 * after the centre update, :78-127); named by the task, so the rule is this library's own and
 * nothing applies it unless asked: components = 8-connected sets of equal labels; per label the
 * largest component stays (ties: the one whose first raster pixel comes first); every other
 * component takes the label of the pixel left of its first raster pixel (above it in column 0)
 * as read from the input map, or keeps its label when that pixel is unlabelled or missing.  One
 * pass; pixels < 0 stay as they are; labels >= nlabels are never relabelled.  labels / out: int32
 * [B][H][W] (different buffers); scratch: imgan_enforce_connectivity_scratch_bytes() bytes. */
size_t imgan_enforce_connectivity_scratch_bytes(int B, int H, int W, int nlabels);
int imgan_enforce_connectivity(const int32_t* labels, int32_t* out, int B, int H, int W, int nlabels,
                               void* scratch, imgan_stream_t stream);

/* ---- dominant colour (mpeg7Descriptors.py:76-130) ----------------------------
 * Segments are given by an int32 label map (ids 0..nseg-1, <0 ignored).
 * centres: float32 [B][nseg][k][3], in/out (initial centres injected by caller).
 * assign: uint8 [B][H][W] cluster index per pixel (0xFF outside any segment).
 * state: per-segment loop state, imgan_dcd_state_bytes() bytes, zero-filled
 * before the first round.  One call = one Lloyd round (:86-107) for every
 * segment that has not met the stop rule yet; first_round != 0 also marks the
 * pixels outside every segment in `assign`.  active_out (device int32[1], may be
 * NULL) receives the number of segments still running.  k <= 16.
 * A cluster that runs empty is re-seeded (:101).  The reference draws the pixel with an
 * unseeded random.sample; here cluster j of a segment takes that segment's pixel p = y*W + x
 * with the smallest (mix32((p * 0x9e3779b1 + seed) ^ ((j + 1) * 0x85ebca6b)), p), mix32 being
 * the murmur3 finaliser: reproducible, independent of the round.  imgan_dcd_segments* follow
 * the same rule with their `seed`. */
size_t imgan_dcd_state_bytes(int B, int nseg, int k);
int imgan_dcd_kmeans_round(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                           int nseg, int k, uint32_t seed, float* centres, uint8_t* assign,
                           void* state, int first_round, int32_t* active_out,
                           imgan_stream_t stream);
/* Per-segment loop state: rounds_out / done_out int32 [B][nseg] (device). */
int imgan_dcd_rounds(const void* state, int B, int nseg, int k, int32_t* rounds_out,
                     int32_t* done_out, imgan_stream_t stream);

/* Device-side, reproducible stand-in for the reference's unseeded random.sample(segment, k)
 * (mpeg7Descriptors.py:79): pixel p competes for slot hash(p, seed) % k of its segment and the
 * smallest (hash, p) wins, giving k distinct pixels per segment.  centres: float32
 * [B][nseg][k][3] out; slots: uint64 [B][nseg][k] scratch (winning hash<<32 | pixel).
 * state: NULL, or the loop state after imgan_dcd_segments (segments done there are skipped). */
int imgan_dcd_init_hash(const uint8_t* lab, const int32_t* labels, int B, int H, int W, int nseg,
                        int k, uint32_t seed, float* centres, uint64_t* slots, const void* state,
                        imgan_stream_t stream);

/* The whole loop of mpeg7Descriptors.py:76-116 for every segment that fits on chip: one
 * workgroup gathers a segment's pixels into LDS, runs its Lloyd rounds there until the
 * reference's stop rule and writes centres, the per-pixel assignment, the per-segment round
 * count (state) and the integer statistics count / sums / sums of squares (stats slots 0-6).
 * use_hash_init != 0 samples the initial centres as imgan_dcd_init_hash does, otherwise
 * `centres` holds them on entry; `seed` also drives the re-seeding of empty clusters (see
 * imgan_dcd_kmeans_round).  Segments too large for LDS are left untouched and counted in
 * n_big_out (device int32): the caller finishes them with imgan_dcd_init_hash /
 * imgan_dcd_kmeans_round, which skip the segments already done.  `state` needs no
 * initialisation; bbox_scratch: imgan_dcd_bbox_bytes() bytes. */
size_t imgan_dcd_bbox_bytes(int B, int nseg);
int imgan_dcd_segments(const uint8_t* lab, const int32_t* labels, int B, int H, int W, int nseg,
                       int k, int use_hash_init, uint32_t seed, float* centres, uint8_t* assign,
                       void* state, uint64_t* stats, void* bbox_scratch, int32_t* n_big_out,
                       int max_rounds, imgan_stream_t stream);

/* The same with the bounding boxes already in place: the first B * nseg * 16 bytes of
 * bbox_scratch hold int32 [B][nseg][4] = xmin, ymin, xmax, ymax of every segment of `labels`
 * (imgan_label_bbox, or imgan_slic_assign_accumulate_bbox when the labels are superpixels),
 * which saves the pass over the label map that finds them (segment_generator's
 * np.where(clusters == seg), mpeg7Descriptors.py:28-33, for every segment at once). */
int imgan_dcd_segments_boxed(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                             int nseg, int k, int use_hash_init, uint32_t seed, float* centres,
                             uint8_t* assign, void* state, uint64_t* stats, void* bbox_scratch,
                             int32_t* n_big_out, int max_rounds, imgan_stream_t stream);

/* The same with bounding boxes AND pixel counts known (bbox int32 [B][nseg][4], counts int32
 * [B][nseg]; imgan_label_bbox / imgan_label_counts, or the last SLIC round and
 * imgan_slic_label_counts): one streaming pass sorts the pixel positions by segment into
 * list_scratch (imgan_dcd_list_bytes() bytes, 4 per pixel), every segment's workgroup then walks
 * its own run (fetching the colours by position) instead of scanning its bounding box, and goes
 * straight to the LDS tier that fits it
 * (np.where(clusters == seg) of mpeg7Descriptors.py:28-33 for all segments in one pass).
 * Same results as imgan_dcd_segments.  Needs W <= 65536 and B*H*W < 2^32.  Counts that
 * disagree with `labels` lose the surplus pixels (never memory safety). */
size_t imgan_dcd_list_bytes(int B, int H, int W, int nseg);
int imgan_dcd_segments_listed(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                              int nseg, int k, int use_hash_init, uint32_t seed, float* centres,
                              uint8_t* assign, void* state, uint64_t* stats, const int32_t* bbox,
                              const int32_t* counts, void* list_scratch, int32_t* n_big_out,
                              int max_rounds, imgan_stream_t stream);

/* counts int32 [B][nseg]: pixels per label (labels outside [0, nseg) are skipped); the
 * np.where(clusters == seg) sizes of mpeg7Descriptors.py:28-33 for every segment at once. */
int imgan_label_counts(const int32_t* labels, int B, int H, int W, int nseg, int32_t* counts,
                       imgan_stream_t stream);

/* mpeg7Descriptors.py:114-129 from the integer statistics: features float32 [B][nseg][k][8] =
 * mean(3), population variance(3), percentage, spatial coherence (16-byte aligned). */
int imgan_dcd_features(const uint64_t* stats, const float* centres, int B, int nseg, int k,
                       float* features, imgan_stream_t stream);

/* mpeg7Descriptors.py:110-129: per (segment, colour) integer statistics.
 * stats: uint64 [B][nseg][k][8] = count, sum c0..c2, sum of squares c0..c2,
 * size of the largest 8-connected component.  scratch: int32 [B][H][W] x 3.  state: the loop
 * state (keeps the sums imgan_dcd_segments already wrote) or NULL (recompute everything). */
int imgan_dcd_stats(const uint8_t* lab, const int32_t* labels, const uint8_t* assign, int B, int H,
                    int W, int nseg, int k, uint64_t* stats, int32_t* scratch, const void* state,
                    imgan_stream_t stream);

/* ---- MPEG-7 colour layout / edge histogram / colour structure ----------------
 * Not present in the reference; defined from ISO/IEC 15938-3,
 * see oracle/mpeg7_std.py.  Whole-image descriptors, one per batch entry.
 * cld_out: int32 [B][3][64] quantised Y/Cb/Cr coefficients in zig-zag order.
 * ehd_out: int32 [B][96] = 80 bin counts + 16 blocks-per-subimage.
 * csd_out: int32 [B][nbins+1] = nbins window counts + number of windows. */
int imgan_cld(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out, void* scratch,
              imgan_stream_t stream);
size_t imgan_cld_scratch_bytes(int B);

/* imgan_cld for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs, outputs and
 * results as imgan_cld; the block sums that imgan_cld keeps in the caller's `scratch` live in a
 * static device array of the library instead (2 KiB per image, 2 MiB per GPU, held until the
 * process ends), so B <= 1024 (IMGAN_E_RANGE beyond).  Calls no hipMalloc / hipFree or any other
 * allocating routine.  Every call on a GPU shares that GPU's array (cleared at the start of
 * each call): calls that overlap on one GPU -- two streams, two threads -- race.
 * The array used is the CURRENT device's (hipSetDevice), which must be the device `stream`
 * belongs to (IMGAN_E_INVALID otherwise).  The first call on a device looks the array's address up
 * (hipGetSymbolAddress, which may load the code object: not a stream operation), so under graph
 * capture call imgan_cld_static_prepare() -- the same look-up and nothing else, for the current
 * device -- once per device before the capture begins; it has no counterpart in the reference
 * either. */
int imgan_cld_static(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out,
                     imgan_stream_t stream);
int imgan_cld_static_prepare(void);
int imgan_ehd(const uint8_t* bgr, int B, int H, int W, int32_t* ehd_out, void* scratch,
              imgan_stream_t stream);
size_t imgan_ehd_scratch_bytes(int B, int H, int W);
/* imgan_cld and imgan_ehd in one call that reads the image once for both sets of sums (each of
 * the two is bound by reading the image); same outputs bit for bit, same scratch buffers
 * (cld_scratch: imgan_cld_scratch_bytes, ehd_scratch: imgan_ehd_scratch_bytes).  No counterpart in
 * the reference (neither descriptor is in it); the batch pipeline's form of the two. */
int imgan_cld_ehd(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out, int32_t* ehd_out,
                  void* cld_scratch, void* ehd_scratch, imgan_stream_t stream);
int imgan_csd(const uint8_t* bgr, int B, int H, int W, int nbins, int32_t* csd_out,
              imgan_stream_t stream);
int imgan_hmmd_quantise(const uint8_t* bgr, uint8_t* out, int64_t npix, int nbins,
                        imgan_stream_t stream);

/* The same three descriptors for every region of a label map (ids 0..nseg-1), one pass over image
 * and labels (BASELINE.json configs[4]: descriptors per superpixel region; not in the reference,
 * defined in oracle/mpeg7_std.py "per region": everything relative to the region's bounding box,
 * only the region's pixels count, empty colour-layout blocks take the region's mean).
 * bbox: int32 [B][nseg][4] = xmin, ymin, xmax, ymax as imgan_label_bbox or
 * imgan_slic_assign_accumulate_bbox leave them.  cld_out int32 [B][nseg][3][64], ehd_out int32
 * [B][nseg][96], csd_out int32 [B][nseg][nbins+1].  A region whose box has more than 24,576 cells
 * does not fit on chip: its rows are zeroed and it is counted in too_big_out (device int32); such
 * regions (merged regions of the class API, up to the whole frame) go through
 * imgan_region_descriptors_large, one region per call.
 * One image holds at most 2^32 / 3 pixels (32-bit offsets inside an image; larger: error).
 * scratch: imgan_region_descriptors_scratch_bytes(B, nseg) bytes. */
size_t imgan_region_descriptors_scratch_bytes(int B, int nseg);
int imgan_region_descriptors(const uint8_t* bgr, const int32_t* labels, const int32_t* bbox, int B,
                             int H, int W, int nseg, int nbins, int32_t* cld_out, int32_t* ehd_out,
                             int32_t* csd_out, int32_t* too_big_out, void* scratch,
                             imgan_stream_t stream);

/* imgan_region_descriptors with the regions' pixels taken from the pixel lists that
 * imgan_dcd_segments_listed left in list_scratch for the SAME label map, boxes and shape (every
 * region's positions as one contiguous run) instead of from a scan of every bounding box against
 * the label map: a superpixel's box holds about 2.2 cells per pixel of its own.  Same outputs,
 * same limits, bit for bit (no counterpart in the reference: it has no per-region form at all).
 * The lists hold 16-bit positions: W <= 65536, H <= 65535, B * H * W < 2^32. */
int imgan_region_descriptors_listed(const uint8_t* bgr, const int32_t* bbox, const void* list_scratch,
                                    int B, int H, int W, int nseg, int nbins, int32_t* cld_out,
                                    int32_t* ehd_out, int32_t* csd_out, int32_t* too_big_out,
                                    void* scratch, imgan_stream_t stream);

/* The same three descriptors of ONE region of ONE image whose bounding box is too large for the
 * on-chip pass above (MPEG7Descriptors is built on the merged regions, optigen_utils.py:46-49, whose
 * boxes reach the whole frame): streaming kernels over the box -- colour-layout block sums and the
 * edge histogram's quadrant sums by run-aggregated global atomics, the colour structure by the
 * whole-image kernel with the region as a mask (sub-sampling from the box's size).  Same
 * definitions (oracle/mpeg7_std.py "per region"), same results.  bgr / labels point at the image
 * (H x W x 3 u8) and its label map; region: the label; x0..y1: its bounding box, inclusive, inside
 * the image (IMGAN_E_INVALID otherwise).  cld_out int32 [3][64], ehd_out int32 [96], csd_out int32
 * [nbins+1].  scratch: imgan_region_descriptors_large_scratch_bytes(H, W) bytes. */
size_t imgan_region_descriptors_large_scratch_bytes(int H, int W);
int imgan_region_descriptors_large(const uint8_t* bgr, const int32_t* labels, int H, int W, int region,
                                   int x0, int y0, int x1, int y1, int nbins, int32_t* cld_out,
                                   int32_t* ehd_out, int32_t* csd_out, void* scratch,
                                   imgan_stream_t stream);

/* ---- spatial relations (spatialextract.py) ------------------------------------
 * imgan_label_bbox: bounding box of every label (cv2.boundingRect of the segment mask,
 * spatialextract.py:9-11,47).  bbox: int32 [B][nseg][4] = xmin, ymin, xmax, ymax
 * (xmax = -1 for a label without pixels).  Labels outside 0..nseg-1 are ignored. */
int imgan_label_bbox(const int32_t* labels, int B, int H, int W, int nseg, int32_t* bbox,
                     imgan_stream_t stream);

/* spatialextract.py:25-38 and :71-171: label counts inside the compass sectors of a segment's
 * enlarged box.  A sector is given as one or two "pieces" (column ranges); for column x of a
 * piece the rows [lo(x), hi(x)) belong to the sector, each bound being a constant or
 * int(a*x + b) in float64, optionally clamped against the constant.
 *   pieces_i: int32 [npieces][8] = slot, sector (0..7), xa, xb, lo_kind, hi_kind, lo_c, hi_c
 *   pieces_d: double [npieces][4] = lo_a, lo_b, hi_a, hi_b
 *   kind: 0 = c, 1 = int(a*x+b), 2 = max(int(a*x+b), c), 3 = min(int(a*x+b), c)
 * counts: int32 [nslots][8][nlab], zero-filled by the caller; counts[slot][sector][l] receives
 * the number of sector pixels labelled l (labels outside 0..nlab-1 are not counted).
 * max_cols: the widest piece (xb - xa).  oob_flag: device int32, zero-filled by the caller,
 * set to 1 if a piece reaches outside the image (the reference would raise IndexError).
 * labels: int32 [H][W] (one image). */
int imgan_spatial_sectors(const int32_t* labels, int H, int W, int nlab, const int32_t* pieces_i,
                          const double* pieces_d, int npieces, int max_cols, int32_t* counts,
                          int32_t* oob_flag, imgan_stream_t stream);

/* ---- region shape / ART (mpeg7Descriptors.py:144-186, 233-275) -----------------
 * For every segment s in 0..nseg-1 of `labels` (int32 [H][W], one image): the value image is 1
 * on the segment, the grey level where labels < 0 and 0 elsewhere (segment_generator :28-38
 * followed by :264-266); cart2radial (:156-185, window means with Python slice rules) gives the
 * polar image and art_transform (:239-256) the 3 x 12 magnitudes, normalised by [0][0].
 * cos_tab / sin_tab: double [H] = cos / sin of y * 2 pi / H (device; computed by the caller so
 * that the sample positions are the reference's bit for bit); r_scale = sqrt(W^2+H^2)/2/W (:153-154).
 * polar_out: uint8 [nseg][H][W] or NULL; art_out: double [nseg][36] (n major);
 * scratch: imgan_region_shape_scratch_bytes(H, W, nseg) bytes.
 * nseg = 1 with labels all < 0 transforms `grey` itself (art_transform of a plain image). */
size_t imgan_region_shape_scratch_bytes(int H, int W, int nseg);
int imgan_region_shape(const uint8_t* grey, const int32_t* labels, int H, int W, int nseg,
                       const double* cos_tab, const double* sin_tab, double r_scale,
                       uint8_t* polar_out, double* art_out, void* scratch, imgan_stream_t stream);

/* ---- homogeneous texture (mpeg7Descriptors.py:277-365, const_data.py) -----------
 * For every segment s in 0..nseg-1 of `labels` (int32 [H][W], one image): the value image is the
 * grey level on the segment and where labels < 0, 0 on other segments (segment_generator
 * :28-38), as float32(g)/255 zero-padded to N x N, N = max(H, W) (:311-319).  Computes the
 * 329-angle Radon transform through scipy.ndimage.rotate's cubic-spline algorithm (:323-331),
 * the shifted row FFT (:334-341) and, for the 30 Gabor channels, the complex sums
 * sum_p = sum ((g * radius) * F)^2 and sum_q = sum (((g * radius) * F)^2 - sum_p)^2 (:351-355).
 *   rot:      double [329][6] = m00, m01, m10, m11, off0, off1: scipy's rotation matrix
 *             [[c, s], [-s, c]] (c, s = cosdg / sindg of rad2deg(pi * per / 329)) and offset
 *   chan:     double [30][4] = centre frequency, 2 * rad_std^2, centre angle (rad), 2 * ang_std^2
 *   angle_p:  double [329] = 2 pi per / 329;  radius_p: double [N] = linspace(0, 1, N)
 *   out:      double [nseg][122] = per channel sum_p (re, im), sum_q (re, im); then the mean and
 *             the population standard deviation of the padded image (:361-362)
 *   scratch:  imgan_texture_scratch_bytes(H, W, nseg) bytes
 * All table pointers are device pointers.  nseg = 1 with labels all < 0 = a plain image. */
size_t imgan_texture_scratch_bytes(int H, int W, int nseg);
int imgan_homogeneous_texture(const uint8_t* grey, const int32_t* labels, int H, int W, int nseg,
                              const double* rot, const double* chan, const double* angle_p,
                              const double* radius_p, double* out, void* scratch,
                              imgan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMGAN_H */
