/* decode_file.c -- a C caller of the drop-in API: the call sequence of the reference's example tool
 * (example/example_tool.c:101-176: startup, parse_header, get_buffer_size, transfer, decode, cleanup) with
 * the HIP runtime in place of the CUDA one. Writes every component plane as a binary PGM, and, with
 * --rgb, an interleaved PPM made on the device by jpeggpu_ext_planes_to_rgbi.
 *
 *   cc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/decode_file.c \
 *      -Ljpeggpu_amd/lib -ljpeggpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/jpeggpu_amd/lib -o decode_file
 *   ./decode_file in.jpg out_prefix [--rgb]
 */
#include <hip/hip_runtime_api.h>
#include <jpeggpu/jpeggpu.h>
#include <jpeggpu/jpeggpu_ext.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
            return EXIT_FAILURE;                                                                \
        }                                                                                       \
    } while (0)
#define CHECK_JPEGGPU(call)                                                                     \
    do {                                                                                        \
        enum jpeggpu_status s_ = (call);                                                        \
        if (s_ != JPEGGPU_SUCCESS) {                                                            \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, jpeggpu_get_status_string(s_)); \
            return EXIT_FAILURE;                                                                \
        }                                                                                       \
    } while (0)

static int write_pnm(const char* path, const char* magic, const uint8_t* px, int w, int h, int channels)
{
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "%s\n%d %d\n255\n", magic, w, h);
    const size_t n = (size_t)w * h * channels;
    const int ok   = fwrite(px, 1, n, f) == n;
    fclose(f);
    return ok ? 0 : -1;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.jpg out_prefix [--rgb]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const int want_rgb = argc > 3 && strcmp(argv[3], "--rgb") == 0;

    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return EXIT_FAILURE;
    }
    fseek(f, 0, SEEK_END);
    const long file_size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* data = NULL; /* pinned, as the header asks for (reference jpeggpu.h:83) */
    CHECK_HIP(hipHostMalloc((void**)&data, (size_t)file_size, hipHostMallocDefault));
    if (fread(data, 1, (size_t)file_size, f) != (size_t)file_size) return EXIT_FAILURE;
    fclose(f);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    jpeggpu_decoder_t decoder;
    CHECK_JPEGGPU(jpeggpu_decoder_startup(&decoder));
    struct jpeggpu_img_info info;
    CHECK_JPEGGPU(jpeggpu_decoder_parse_header(decoder, &info, data, (size_t)file_size));

    size_t tmp_size = 0;
    CHECK_JPEGGPU(jpeggpu_decoder_get_buffer_size(decoder, &tmp_size));
    void* d_tmp = NULL;
    CHECK_HIP(hipMalloc(&d_tmp, tmp_size));

    struct jpeggpu_img img;
    memset(&img, 0, sizeof(img));
    for (int c = 0; c < info.num_components; ++c) {
        CHECK_HIP(hipMalloc((void**)&img.image[c], (size_t)info.sizes_x[c] * info.sizes_y[c]));
        img.pitch[c] = info.sizes_x[c];
    }

    CHECK_JPEGGPU(jpeggpu_decoder_transfer(decoder, d_tmp, tmp_size, stream));
    CHECK_JPEGGPU(jpeggpu_decoder_decode(decoder, &img, d_tmp, tmp_size, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    char path[4096];
    for (int c = 0; c < info.num_components; ++c) {
        const size_t n = (size_t)info.sizes_x[c] * info.sizes_y[c];
        uint8_t* h     = (uint8_t*)malloc(n);
        CHECK_HIP(hipMemcpy(h, img.image[c], n, hipMemcpyDeviceToHost));
        snprintf(path, sizeof(path), "%s_%d.pgm", argv[2], c);
        if (write_pnm(path, "P5", h, info.sizes_x[c], info.sizes_y[c], 1)) return EXIT_FAILURE;
        free(h);
    }
    if (want_rgb) {
        /* full resolution = the size of the component with the largest sampling factors */
        int w = 0, h = 0;
        for (int c = 0; c < info.num_components; ++c) {
            if (info.sizes_x[c] > w) w = info.sizes_x[c];
            if (info.sizes_y[c] > h) h = info.sizes_y[c];
        }
        uint8_t* d_rgb = NULL;
        CHECK_HIP(hipMalloc((void**)&d_rgb, (size_t)w * h * 3));
        CHECK_JPEGGPU(jpeggpu_ext_planes_to_rgbi(&info, &img, d_rgb, 3 * w, w, h, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        uint8_t* rgb = (uint8_t*)malloc((size_t)w * h * 3);
        CHECK_HIP(hipMemcpy(rgb, d_rgb, (size_t)w * h * 3, hipMemcpyDeviceToHost));
        snprintf(path, sizeof(path), "%s.ppm", argv[2]);
        if (write_pnm(path, "P6", rgb, w, h, 3)) return EXIT_FAILURE;
        free(rgb);
        CHECK_HIP(hipFree(d_rgb));
    }

    for (int c = 0; c < info.num_components; ++c) CHECK_HIP(hipFree(img.image[c]));
    CHECK_HIP(hipFree(d_tmp));
    CHECK_JPEGGPU(jpeggpu_decoder_cleanup(decoder));
    CHECK_HIP(hipStreamDestroy(stream));
    CHECK_HIP(hipHostFree(data));
    printf("%d components, %dx%d\n", info.num_components, info.sizes_x[0], info.sizes_y[0]);
    return EXIT_SUCCESS;
}
