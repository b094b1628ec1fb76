// Host build of vartrix_amd/csrc/vtx_inflate_core.h — the per-lane DEFLATE decoder of bgzf_inflate_kernel — for the CPU unit tests
// (tests/test_inflate_core.py: against zlib on every block kind, bit flips, the reference's BAM).  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../vartrix_amd/csrc/vtx_inflate_core.h"

extern "C" {

// Returns the decoder's status (0 = accepted, out holds out_len bytes).  The input is copied into a padded buffer (the device buffer
// is padded the same way); `stride` > 1 lays the lane's scratch out interleaved, as the kernel's LDS is.
uint32_t vtxt_inflate(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, int stride, uint32_t* trips) {
    std::vector<uint8_t> padded((size_t)in_len + 16, 0xa5);
    if (in_len) memcpy(padded.data(), in, in_len);
    if (stride < 1) stride = 1;
    std::vector<uint8_t> sb((size_t)vtxi::BYTES * stride, 0xde);
    std::vector<uint32_t> sh((size_t)vtxi::HI_WORDS * stride, 0xdeadbeefu);
    std::vector<uint16_t> scn((size_t)vtxi::CNT_WORDS * stride, 0xdead);
    const vtxi::Scratch sc{sb.data(), sh.data(), scn.data(), stride};
    return vtxi::inflate_block(padded.data(), in_len, out, out_len, sc, trips);
}

}
