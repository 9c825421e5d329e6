// Entry points declared in include/imgan.h that are not implemented yet.
#include "common.h"
#include "../../include/imgan.h"
#define NOT_YET(name) return imgan::fail(imgan::E_INVALID, #name ": not implemented yet")
extern "C" {
int imgan_median7_u8c3(const uint8_t*, uint8_t*, int, int, int, imgan_stream_t) { NOT_YET(imgan_median7_u8c3); }
int imgan_rag_accumulate(const uint8_t*, const int32_t*, int, int, int, int, uint64_t*, uint64_t*, int64_t, imgan_stream_t) { NOT_YET(imgan_rag_accumulate); }
int imgan_rag_merge_host(const int32_t*, int, const int32_t*, int64_t, double, int32_t*, int32_t*, int64_t*, int32_t*, int32_t*) { NOT_YET(imgan_rag_merge_host); }
int imgan_relabel_lut(const int32_t*, const int32_t*, int32_t*, int, int64_t, int, imgan_stream_t) { NOT_YET(imgan_relabel_lut); }
int imgan_paint_means(uint8_t*, const int32_t*, const uint8_t*, int, int64_t, int, imgan_stream_t) { NOT_YET(imgan_paint_means); }
int imgan_cc_label(const int32_t*, int32_t*, int, int, int, imgan_stream_t) { NOT_YET(imgan_cc_label); }
size_t imgan_dcd_state_bytes(int, int, int) { return 0; }
int imgan_dcd_kmeans_round(const uint8_t*, const int32_t*, int, int, int, int, int, float*, uint8_t*, void*, int32_t*, imgan_stream_t) { NOT_YET(imgan_dcd_kmeans_round); }
int imgan_dcd_stats(const uint8_t*, const int32_t*, const uint8_t*, int, int, int, int, int, uint64_t*, int32_t*, imgan_stream_t) { NOT_YET(imgan_dcd_stats); }
int imgan_cld(const uint8_t*, int, int, int, int32_t*, void*, imgan_stream_t) { NOT_YET(imgan_cld); }
size_t imgan_cld_scratch_bytes(int) { return 0; }
int imgan_ehd(const uint8_t*, int, int, int, int32_t*, void*, imgan_stream_t) { NOT_YET(imgan_ehd); }
size_t imgan_ehd_scratch_bytes(int, int, int) { return 0; }
int imgan_csd(const uint8_t*, int, int, int, int, int32_t*, imgan_stream_t) { NOT_YET(imgan_csd); }
int imgan_hmmd_quantise(const uint8_t*, uint8_t*, int64_t, int, imgan_stream_t) { NOT_YET(imgan_hmmd_quantise); }
}
