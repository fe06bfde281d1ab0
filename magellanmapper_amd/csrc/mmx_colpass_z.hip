// Z pass instantiations (see mmx_colpass.inc for the kernel and its design notes).
#include "mmx_colpass.inc"

namespace {
template <int R>
int launch_z(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_cols,
             int64_t slot_elems, const mmx_taps_f32& taps, float* d_gz, float* d_gzz, hipStream_t s)
{
    dim3 grid((max_cols + MMX_WG - 1) / MMX_WG, n_blocks);
    const int sy = (int)vol->stride_y, sx = (int)vol->stride_x;
    if (vol->dtype == MMX_U16)
        hipLaunchKernelGGL((zpass_kernel<R, uint16_t>), grid, dim3(MMX_WG), 0, s,
                           (const uint16_t*)vol->d_data, vol->stride_z, sy, sx, d_blocks, slot_elems,
                           d_gz, d_gzz, taps);
    else if (vol->dtype == MMX_F32)
        hipLaunchKernelGGL((zpass_kernel<R, float>), grid, dim3(MMX_WG), 0, s,
                           (const float*)vol->d_data, vol->stride_z, sy, sx, d_blocks, slot_elems,
                           d_gz, d_gzz, taps);
    else if (vol->dtype == MMX_U8)
        hipLaunchKernelGGL((zpass_kernel<R, uint8_t>), grid, dim3(MMX_WG), 0, s,
                           (const uint8_t*)vol->d_data, vol->stride_z, sy, sx, d_blocks, slot_elems,
                           d_gz, d_gzz, taps);
    else
        return MMX_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? MMX_OK : MMX_ERR_HIP;
}
}  // namespace

#define MMX_FOR_EACH_RADIUS(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

int mmx_launch_zpass(const mmx_volume* vol, const mmx_block* d_blocks, int n_blocks, int max_cols,
                     int64_t slot_elems, const mmx_taps_f32& taps, int radius,
                     float* d_gz, float* d_gzz, hipStream_t stream)
{
    switch (radius) {
#define X(R) case R: return launch_z<R>(vol, d_blocks, n_blocks, max_cols, slot_elems, taps, d_gz, d_gzz, stream);
        MMX_FOR_EACH_RADIUS(X)
#undef X
        default: return MMX_ERR_UNSUPPORTED;
    }
}
