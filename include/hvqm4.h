/*
 * hvqm4.h -- HVQM4 1.5 SDK C API, served by the MI355X reconstruction back end.
 *
 * These seven entry points are the drop-in boundary: names from the reference's
 * symbols.inc:2-8, signatures and semantics from h4m_audio_decode.c (cited per function).
 * In the reference they are `static` functions reached by #include-ing the .c file; here
 * they are exported from libhvqm4_amd.so.  INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Ownership (h4m:2409-2419, 2340-2350): the caller allocates and frees everything -- a work
 * buffer of HVQM4BuffSize() bytes and the picture buffers of w*h*(hs*vs+2)/(hs*vs) bytes,
 * planes Y|U|V tightly packed.  `frame` points 4 bytes past the record start (after disp_id,
 * h4m:2100).  Nothing behind the picture's last section is read (the reference reads up to 3
 * bytes past it, h4m:2080-2082): the length comes from the picture's own section table, and
 * HVQM4SetMaxFrameSize bounds the walk over that table (tests/native/sdk_bounds_asan.c).
 * All calls are synchronous: on return `present` holds the decoded picture.  HVQM4DecodePpic
 * reads `present` like the reference does when the picture carries future-referencing
 * macroblocks (h4m:2058-2061): the buffer's content at the call is part of the input then.
 *
 * The functions return void like the reference; failures (no GPU, HIP error, unsupported
 * geometry) are reported on stderr and through HVQM4GetLastError() -- `present` is then left
 * untouched.  There is no CPU fallback.
 */
#ifndef HVQM4_H
#define HVQM4_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sizeof(VideoState) and the offset of its `padding` member in the reference's NATIVE x86-64
 * build (h4m:470-507; probed) -- kept so that a caller written against the
 * reference (`state->padding[0] = is_1_5;` h4m:2414-2417) works unchanged. */
#define HVQM4_VIDEOSTATE_SIZE      28120
#define HVQM4_VIDEOSTATE_PADDING   28097

typedef struct VideoState {
    uint8_t opaque0[HVQM4_VIDEOSTATE_PADDING];
    uint8_t padding[3];            /* padding[0] != 0: stream is HVQM4 1.5 (h4m:2416) */
    uint8_t opaque1[HVQM4_VIDEOSTATE_SIZE - HVQM4_VIDEOSTATE_PADDING - 3];
} VideoState;

typedef struct SeqObj {            /* h4m:516-523 */
    VideoState *state;
    uint16_t width;
    uint16_t height;
    uint8_t h_samp;
    uint8_t v_samp;
} SeqObj;

typedef struct VideoInfo {         /* h4m:533-540 */
    uint16_t hres;
    uint16_t vres;
    uint8_t h_samp;
    uint8_t v_samp;
    uint8_t video_mode;
} VideoInfo;

void     HVQM4InitDecoder(void);                                                  /* h4m:275  */
void     HVQM4InitSeqObj(SeqObj *seqobj, VideoInfo *videoinfo);                   /* h4m:819  */
uint32_t HVQM4BuffSize(SeqObj *seqobj);                                           /* h4m:828  */
void     HVQM4SetBuffer(SeqObj *seqobj, void *workbuff);                          /* h4m:957  */
void     HVQM4DecodeIpic(SeqObj *seqobj, uint8_t const *frame, void *present);    /* h4m:1970 */
void     HVQM4DecodePpic(SeqObj *seqobj, uint8_t const *frame, void *present, void *past);                /* h4m:2058 */
void     HVQM4DecodeBpic(SeqObj *seqobj, uint8_t const *frame, void *present, void *past, void *future);  /* h4m:2018 */

/* ---- additions (not in the SDK) ---- */
int      HVQM4GetLastError(void);            /* 0 = ok; HVQ_E_* otherwise; sticky until read */
const char *HVQM4GetLastErrorString(void);
void     HVQM4SetVersion15(SeqObj *seqobj, int is_1_5);   /* explicit form of the padding[0] hack */
void     HVQM4ReleaseBuffer(SeqObj *seqobj); /* frees the GPU resources bound by HVQM4SetBuffer; call before freeing workbuff */
/* Optional: the number of readable bytes at every `frame` handed to the decode calls (the container's max_frame_size,
 * h4m:2188; a player allocates exactly that).  The SDK signatures carry no length; without this the length of a picture
 * is taken from its own section table (h4m:1978-1993, 2029-2044), whose words the reference reads unchecked too. */
void     HVQM4SetMaxFrameSize(SeqObj *seqobj, uint32_t bytes);

#ifdef __cplusplus
}
#endif
#endif
