/*
 * jpeggpu.h -- C ABI of the MI355X-native baseline JPEG decoder.
 *
 * Drop-in boundary: every name, struct layout and enum value below mirrors the
 * reference's public header so that existing callers only need to swap the
 * stream type (cudaStream_t -> hipStream_t; a `cudaStream_t` alias is offered
 * when JPEGGPU_CUDA_COMPAT_NAMES is defined).
 *
 *   reference interface replaced                     | declared here
 *   -------------------------------------------------+-----------------------------
 *   include/jpeggpu/jpeggpu.h:33  JPEGGPU_MAX_COMP   | JPEGGPU_MAX_COMP
 *   include/jpeggpu/jpeggpu.h:35-36 opaque handle    | jpeggpu_decoder_t
 *   include/jpeggpu/jpeggpu.h:38-52 status enum      | enum jpeggpu_status (values 0..6)
 *   include/jpeggpu/jpeggpu.h:55                     | jpeggpu_get_status_string
 *   include/jpeggpu/jpeggpu.h:59                     | jpeggpu_decoder_startup
 *   include/jpeggpu/jpeggpu.h:62                     | jpeggpu_set_logging
 *   include/jpeggpu/jpeggpu.h:65-68                  | struct jpeggpu_subsampling
 *   include/jpeggpu/jpeggpu.h:70                     | is_css_444
 *   include/jpeggpu/jpeggpu.h:72-79                  | struct jpeggpu_img_info
 *   include/jpeggpu/jpeggpu.h:84-85                  | jpeggpu_decoder_parse_header
 *   include/jpeggpu/jpeggpu.h:88                     | jpeggpu_decoder_get_buffer_size
 *   include/jpeggpu/jpeggpu.h:92-93                  | jpeggpu_decoder_transfer
 *   include/jpeggpu/jpeggpu.h:97-100                 | struct jpeggpu_img
 *   include/jpeggpu/jpeggpu.h:104-109                | jpeggpu_decoder_decode
 *   include/jpeggpu/jpeggpu.h:111                    | jpeggpu_decoder_cleanup
 *
 * Streams are passed as an opaque pointer-sized handle (`jpeggpu_stream_t`) so
 * that FFI callers (ctypes, cgo, JNI) do not need the HIP headers. When
 * <hip/hip_runtime.h> has been included before this header the handle is the
 * real `hipStream_t`; both are the same pointer type at the ABI level.
 */
#ifndef JPEGGPU_JPEGGPU_H_
#define JPEGGPU_JPEGGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(HIP_INCLUDE_HIP_HIP_RUNTIME_H) || defined(HIP_INCLUDE_HIP_HIP_RUNTIME_API_H)
typedef hipStream_t jpeggpu_stream_t;
#else
typedef struct ihipStream_t* jpeggpu_stream_t;
#endif
#ifdef JPEGGPU_CUDA_COMPAT_NAMES
typedef jpeggpu_stream_t cudaStream_t;
#endif

#define JPEGGPU_MAX_COMP 4

struct jpeggpu_decoder;
typedef struct jpeggpu_decoder* jpeggpu_decoder_t;

enum jpeggpu_status {
    JPEGGPU_SUCCESS = 0,
    /* The user provided an illegal argument to a function. */
    JPEGGPU_INVALID_ARGUMENT = 1,
    /* The JPEG stream is not compatible with the specification. */
    JPEGGPU_INVALID_JPEG = 2,
    /* An error inside the library (or the HIP runtime) occurred. */
    JPEGGPU_INTERNAL_ERROR = 3,
    /* The JPEG stream is valid but uses a feature outside baseline 8-bit Huffman. */
    JPEGGPU_NOT_SUPPORTED = 4,
    /* The system is out of host memory. */
    JPEGGPU_OUT_OF_HOST_MEMORY = 5,
    /* The JPEG stream is invalid, likely due to being incomplete. */
    JPEGGPU_INCOMPLETE_BITSTREAM = 6
};

/* Description of a status code (static storage). */
const char* jpeggpu_get_status_string(enum jpeggpu_status stat);

/* If JPEGGPU_SUCCESS is returned, jpeggpu_decoder_cleanup must be called before
 * the program ends, whatever the intermediate calls return. The decoder owns
 * host memory only; all device memory is owned by the caller. */
enum jpeggpu_status jpeggpu_decoder_startup(jpeggpu_decoder_t* decoder);

/* Logging to stdout, off by default. */
enum jpeggpu_status jpeggpu_set_logging(jpeggpu_decoder_t decoder, int do_logging);

/* Sampling factors as found in the frame header, each in [1, 4]. */
struct jpeggpu_subsampling {
    int x[JPEGGPU_MAX_COMP];
    int y[JPEGGPU_MAX_COMP];
};

int is_css_444(struct jpeggpu_subsampling css, int num_components);

struct jpeggpu_img_info {
    /* Plane sizes ceil(W*h_c/h_max) x ceil(H*v_c/v_max); NOT rounded to the MCU. */
    int sizes_x[JPEGGPU_MAX_COMP];
    int sizes_y[JPEGGPU_MAX_COMP];
    int num_components;
    struct jpeggpu_subsampling subsampling;
};

/* Parse markers, tables and the restart-segment structure. Host only, no GPU
 * work. `data` is borrowed until the copy enqueued by jpeggpu_decoder_transfer
 * has executed; it should be page-locked (hipHostMalloc) for a truly async copy. */
enum jpeggpu_status jpeggpu_decoder_parse_header(
    jpeggpu_decoder_t decoder, struct jpeggpu_img_info* img_info, const uint8_t* data, size_t size);

/* Size in bytes of the temporary device buffer needed for the parsed image. */
enum jpeggpu_status jpeggpu_decoder_get_buffer_size(jpeggpu_decoder_t decoder, size_t* tmp_size);

/* Enqueue the host-to-device copies (entropy-coded bytes + one table blob).
 * d_tmp must be 256-byte aligned device memory of at least tmp_size bytes. */
enum jpeggpu_status jpeggpu_decoder_transfer(
    jpeggpu_decoder_t decoder, void* d_tmp, size_t tmp_size, jpeggpu_stream_t stream);

/* Device output image: one plane per component at its native (possibly
 * subsampled) resolution, `pitch[c]` bytes between rows. */
struct jpeggpu_img {
    uint8_t* image[JPEGGPU_MAX_COMP];
    int pitch[JPEGGPU_MAX_COMP];
};

/* Enqueue the GPU decode on `stream`; d_tmp/tmp_size must be those given to
 * jpeggpu_decoder_transfer. Nothing blocks the host. */
enum jpeggpu_status jpeggpu_decoder_decode(
    jpeggpu_decoder_t decoder,
    struct jpeggpu_img* img,
    void* d_tmp,
    size_t tmp_size,
    jpeggpu_stream_t stream);

enum jpeggpu_status jpeggpu_decoder_cleanup(jpeggpu_decoder_t decoder);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* JPEGGPU_JPEGGPU_H_ */
