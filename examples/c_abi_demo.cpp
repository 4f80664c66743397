// Torch-free use of the C ABI (include/giga_hip.h): a plain HIP host program that links libgiga_hip.so, packs a flat
// parameter file, and runs the encoder + the four decoder heads on raw device buffers.  It is what a non-Python host
// (a C++ planner, a ROS node) would write; tests/test_c_abi_demo.py feeds it the same inputs as the Python path and
// compares the outputs byte streams.
//
//   c_abi_demo <params.bin> <tsdf.bin> <points.bin> <B> <N> <out.bin>
//     params.bin : giga_param_count(15) float32, reference state-dict order
//     tsdf.bin   : B*40*40*40 float32      points.bin : B*N*3 float32
//     out.bin    : qual[B*N] rot[B*N*4] width[B*N] occ[B*N] float32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/giga_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GIGA(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s: %s (%d)\n", #x, giga_strerror(r_), r_); return 3; } } while (0)

static bool read_file(const char* path, std::vector<float>& v, size_t n) {
    v.resize(n);
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t got = std::fread(v.data(), sizeof(float), n, f);
    std::fclose(f);
    return got == n;
}

int main(int argc, char** argv) {
    if (argc != 7) { std::fprintf(stderr, "usage: %s params.bin tsdf.bin points.bin B N out.bin\n", argv[0]); return 1; }
    const int B = std::atoi(argv[4]), N = std::atoi(argv[5]);
    const int heads = 15;                                  // qual | rot | width | occupancy
    if (giga_abi_version() != 3) { std::fprintf(stderr, "ABI version mismatch\n"); return 1; }
    std::vector<float> params, tsdf, pts;
    const size_t nparam = giga_param_count(heads);
    if (!read_file(argv[1], params, nparam) || !read_file(argv[2], tsdf, (size_t)B * 64000) ||
        !read_file(argv[3], pts, (size_t)B * N * 3)) { std::fprintf(stderr, "short input file\n"); return 1; }

    std::vector<unsigned char> blob(giga_packed_bytes());
    CHECK_GIGA(giga_pack_weights(params.data(), nparam, heads, blob.data(), blob.size()));

    const size_t P = (size_t)B * N, ws_bytes = giga_encoder_workspace_bytes(B, 0);
    void *d_blob, *d_ws, *d_planes;
    float *d_tsdf, *d_pts, *d_out;
    CHECK_HIP(hipMalloc(&d_blob, blob.size()));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMalloc(&d_planes, (size_t)3 * B * 40 * 40 * 32 * sizeof(float)));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_tsdf), tsdf.size() * sizeof(float)));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_pts), pts.size() * sizeof(float)));
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_out), P * 7 * sizeof(float)));
    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    CHECK_HIP(hipMemcpyAsync(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
    CHECK_HIP(hipMemcpyAsync(d_tsdf, tsdf.data(), tsdf.size() * sizeof(float), hipMemcpyHostToDevice, s));
    CHECK_HIP(hipMemcpyAsync(d_pts, pts.data(), pts.size() * sizeof(float), hipMemcpyHostToDevice, s));

    float *qual = d_out, *rot = d_out + P, *width = d_out + 5 * P, *occ = d_out + 6 * P;
    // encoder (fp32) -> NHWC planes; then all four heads in one fused launch (post = 1: sigmoid(qual), normalised rot)
    CHECK_GIGA(giga_encoder_forward(d_tsdf, d_blob, d_planes, nullptr, B, 0, d_ws, ws_bytes, s));
    CHECK_GIGA(giga_decoder_forward(d_planes, d_pts, d_blob, heads, qual, rot, width, occ, B, N, 0, 1, s));

    std::vector<float> out(P * 7);
    CHECK_HIP(hipMemcpyAsync(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    CHECK_HIP(hipStreamSynchronize(s));
    FILE* f = std::fopen(argv[6], "wb");
    if (!f || std::fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { std::fprintf(stderr, "cannot write output\n"); return 1; }
    std::fclose(f);
    std::printf("c_abi_demo: B=%d N=%d qual[0]=%.6f occ[0]=%.6f\n", B, N, out[0], out[6 * P]);
    return 0;
}
