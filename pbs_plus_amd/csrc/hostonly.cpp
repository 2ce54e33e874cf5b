// libpbsgpu host side, part 4: the entry points that need neither HIP nor a device — configuration,
// status text, dynamic-index parsing, payload-stream sizing. Kept in their own translation unit so the
// CPU test-suite can build them (with reuse.cpp) under AddressSanitizer/UBSan (tests/native/).
#include <cstring>

#include "../../include/pbsgpu.h"

namespace {

// casync / Proxmox BUZHASH_TABLE. EXTERNAL and UNPINNED: recalled, not copied out of any file in the reference tree (the
// parity checker carries its own copy); what corroborates it is the table's designed balance (exactly 128 one-bits per bit column, checked in
// tests/test_oracle_buzhash.py) and the two hash-determined chunk sizes recalled from upstream's tests. The table is an
// INPUT everywhere (pbsgpu_config_init(avg, table, ...)): inject the module's own constant when wiring this in.
const uint32_t kDefaultTable[256] = {
    0x458be752, 0xc10748cc, 0xfbbcdbb8, 0x6ded5b68, 0xb10a82b5, 0x20d75648, 0xdfc5665f, 0xa8428801,
    0x7ebf5191, 0x841135c7, 0x65cc53b3, 0x280a597c, 0x16f60255, 0xc78cbc3e, 0x294415f5, 0xb938d494,
    0xec85c4e6, 0xb7d33edc, 0xe549b544, 0xfdeda5aa, 0x882bf287, 0x3116737c, 0x05569956, 0xe8cc1f68,
    0x0806ac5e, 0x22a14443, 0x15297e10, 0x50d090e7, 0x4ba60f6f, 0xefd9f1a7, 0x5c5c885c, 0x82482f93,
    0x9bfd7c64, 0x0b3e7276, 0xf2688e77, 0x8fad8abc, 0xb0509568, 0xf1ada29f, 0xa53efdfe, 0xcb2b1d00,
    0xf2a9e986, 0x6463432b, 0x95094051, 0x5a223ad2, 0x9be8401b, 0x61e579cb, 0x1a556a14, 0x5840fdc2,
    0x9261ddf6, 0xcde002bb, 0x52432bb0, 0xbf17373e, 0x7b7c222f, 0x2955ed16, 0x9f10ca59, 0xe840c4c9,
    0xccabd806, 0x14543f34, 0x1462417a, 0x0d4a1f9c, 0x087ed925, 0xd7f8f24c, 0x7338c425, 0xcf86c8f5,
    0xb19165cd, 0x9891c393, 0x325384ac, 0x0308459d, 0x86141d7e, 0xc922116a, 0xe2ffa6b6, 0x53f52aed,
    0x2cd86197, 0xf5b9f498, 0xbf319c8f, 0xe0411fae, 0x977eb18c, 0xd8770976, 0x9833466a, 0xc674df7f,
    0x8c297d45, 0x8ca48d26, 0xc49ed8e2, 0x7344f874, 0x556f79c7, 0x6b25eaed, 0xa03e2b42, 0xf68f66a4,
    0x8e8b09a2, 0xf2e0e62a, 0x0d3a9806, 0x9729e493, 0x8c72b0fc, 0x160b94f6, 0x450e4d3d, 0x7a320e85,
    0xbef8f0e1, 0x21d73653, 0x4e3d977a, 0x1e7b3929, 0x1cc6c719, 0xbe478d53, 0x8d752809, 0xe6d8c2c6,
    0x275f0892, 0xc8acc273, 0x4cc21580, 0xecc4a617, 0xf5f7be70, 0xe795248a, 0x375a2fe9, 0x425570b6,
    0x8898dcf8, 0xdc2d97c4, 0x0106114b, 0x364dc22f, 0x1e0cad1f, 0xbe63803c, 0x5f69fac2, 0x4d5afa6f,
    0x1bc0dfb5, 0xfb273589, 0x0ea47f7b, 0x3c1c2b50, 0x21b2a932, 0x6b1223fd, 0x2fe706a8, 0xf9bd6ce2,
    0xa268e64e, 0xe987f486, 0x3eacf563, 0x1ca2018c, 0x65e18228, 0x2207360a, 0x57cf1715, 0x34c37d2b,
    0x1f8f3cde, 0x93b657cf, 0x31a019fd, 0xe69eb729, 0x8bca7b9b, 0x4c9d5bed, 0x277ebeaf, 0xe0d8f8ae,
    0xd150821c, 0x31381871, 0xafc3f1b0, 0x927db328, 0xe95effac, 0x305a47bd, 0x426ba35b, 0x1233af3f,
    0x686a5b83, 0x50e072e5, 0xd9d3bb2a, 0x8befc475, 0x487f0de6, 0xc88dff89, 0xbd664d5e, 0x971b5d18,
    0x63b14847, 0xd7d3c1ce, 0x7f583cf3, 0x72cbcb09, 0xc0d0a81c, 0x7fa3429b, 0xe9158a1b, 0x225ea19a,
    0xd8ca9ea3, 0xc763b282, 0xbb0c6341, 0x020b8293, 0xd4cd299d, 0x58cfa7f8, 0x91b4ee53, 0x37e4d140,
    0x95ec764c, 0x30f76b06, 0x5ee68d24, 0x679c8661, 0xa41979c2, 0xf2b61284, 0x4fac1475, 0x0adb49f9,
    0x19727a23, 0x15a7e374, 0xc43a18d5, 0x3fb1aa73, 0x342fc615, 0x924c0793, 0xbee2d7f0, 0x8a279de9,
    0x4aa2d70c, 0xe24dd37f, 0xbe862c0b, 0x177c22c2, 0x5388e5ee, 0xcd8a7510, 0xf901b4fd, 0xdbc13dbc,
    0x6c0bae5b, 0x64efe8c7, 0x48b02079, 0x80331a49, 0xca3d8ae6, 0xf3546190, 0xfed7108b, 0xc49b941b,
    0x32baf4a9, 0xeb833a4a, 0x88a3f1a5, 0x3a91ce0a, 0x3cc27da1, 0x7112e684, 0x4a3096b1, 0x3794574c,
    0xa3c8b6f3, 0x1d213941, 0x6e0a2e00, 0x233479f1, 0x0f4cd82f, 0x6093edd2, 0x5d7d209e, 0x464fe319,
    0xd4dcac9e, 0x0db845cb, 0xfb5e4bc3, 0xe0256ce1, 0x09fb4ed1, 0x0914be1e, 0xa5bdb2c3, 0xc6eb57bb,
    0x30320350, 0x3f397e91, 0xa67791bc, 0x86bc0e2c, 0xefa0a7e2, 0xe9ff7543, 0xe733612c, 0xd185897b,
    0x329e5388, 0x91dd236b, 0x2ecb0d93, 0xf4d82a3d, 0x35b5c03f, 0xe4e606f0, 0x05b21843, 0x37b45964,
    0x5eff22f4, 0x6027f4cc, 0x77178b3c, 0xae507131, 0x7bf7cabc, 0xf9c18d66, 0x593ade65, 0xd95ddf11,
};

}  // namespace

extern "C" {

extern const uint8_t pbsgpu_didx_magic[8];
// DYNAMIC_SIZED_CHUNK_INDEX_1_0 of the Proxmox Backup file formats. EXTERNAL and UNPINNED: typed from memory of the
// published format, no .didx fixture exists under the reference tree to check it against; a
// maintainer must confirm it (and the 4096-byte header layout in stream.cpp) against datastore.ParseDynamicIndex.
const uint8_t pbsgpu_didx_magic[8] = {28, 145, 78, 165, 25, 186, 179, 205};

const char *pbsgpu_strerror(int status) {
    switch (status) {
    case PBSGPU_OK: return "ok";
    case PBSGPU_E_INVALID: return "invalid argument";
    case PBSGPU_E_NO_DEVICE: return "no usable HIP device";
    case PBSGPU_E_HIP: return "HIP runtime error";
    case PBSGPU_E_NOMEM: return "out of memory";
    case PBSGPU_E_CAPACITY: return "output buffer too small";
    case PBSGPU_E_BUSY: return "all in-flight slots busy";
    case PBSGPU_E_TICKET: return "unknown ticket";
    case PBSGPU_E_STATE: return "invalid state";
    default: return "unknown status";
    }
}

int pbsgpu_abi_version(void) { return PBSGPU_ABI_VERSION; }
const uint32_t *pbsgpu_default_table(void) { return kDefaultTable; }


int pbsgpu_config_init(uint64_t avg, const uint32_t *table, pbsgpu_config *out) {
    if (!out) return PBSGPU_E_INVALID;
    if (avg < 256 || avg > (1ull << 28) || (avg & (avg - 1)) != 0) return PBSGPU_E_INVALID;
    out->avg = (uint32_t)avg;
    out->min = (uint32_t)(avg >> 2);
    out->max = (uint32_t)(avg << 2);
    out->window = 64u;
    out->mask = (uint32_t)(avg * 2 - 1);
    out->break_min = out->mask - 2;
    std::memcpy(out->table, table ? table : kDefaultTable, sizeof(out->table));
    return PBSGPU_OK;
}

int pbsgpu_didx_size(uint64_t nrecords, uint64_t *nbytes) {
    if (!nbytes) return PBSGPU_E_INVALID;
    *nbytes = PBSGPU_DIDX_HEADER_SIZE + nrecords * 40;
    return PBSGPU_OK;
}

int pbsgpu_didx_decode(const uint8_t *in, uint64_t nbytes, pbsgpu_record *out, uint64_t cap, uint64_t *n,
                       int64_t *ctime, uint8_t index_csum[32]) {
    if (!in || !n) return PBSGPU_E_INVALID;
    if (nbytes < PBSGPU_DIDX_HEADER_SIZE || std::memcmp(in, pbsgpu_didx_magic, 8) != 0) return PBSGPU_E_INVALID;
    const uint64_t body = nbytes - PBSGPU_DIDX_HEADER_SIZE;
    if (body % 40) return PBSGPU_E_INVALID;
    const uint64_t cnt = body / 40;
    *n = cnt;
    if (ctime) {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v |= (uint64_t)in[24 + i] << (8 * i);
        *ctime = (int64_t)v;
    }
    if (index_csum) std::memcpy(index_csum, in + 32, 32);
    if (cnt > cap || (!out && cnt)) return PBSGPU_E_CAPACITY;
    const uint8_t *ent = in + PBSGPU_DIDX_HEADER_SIZE;
    uint64_t prev = 0;
    for (uint64_t i = 0; i < cnt; ++i) {
        uint64_t end = 0;
        for (int b = 0; b < 8; ++b) end |= (uint64_t)ent[i * 40 + b] << (8 * b);
        if (end < prev || end - prev > 0xffffffffull) return PBSGPU_E_INVALID;
        out[i].end = end;
        std::memcpy(out[i].digest, ent + i * 40 + 8, 32);
        out[i].segment = 0;
        out[i].size = (uint32_t)(end - prev);
        prev = end;
    }
    return PBSGPU_OK;
}

int pbsgpu_payload_format_default(pbsgpu_payload_format *out) {
    if (!out) return PBSGPU_E_INVALID;
    // pxar v2 type constants: EXTERNAL and UNPINNED (typed from memory of the published format; no fixture in the
    // reference tree pins them). Injectable through pbsgpu_payload_format; a maintainer must confirm them against the
    // module's format package.
    out->payload_type = 0x28147a1b0b7c1a25ull;  // PXAR_PAYLOAD
    out->start_type = 0x834c68c2194a4ed2ull;    // PXAR_PAYLOAD_START_MARKER
    out->tail_type = 0x6c72b78b984c81b5ull;     // PXAR_PAYLOAD_TAIL_MARKER
    out->with_start = 1;
    out->with_tail = 1;
    return PBSGPU_OK;
}

int pbsgpu_payload_size(const pbsgpu_segment *files, uint32_t nfiles, const pbsgpu_payload_format *fmt,
                        uint64_t *nbytes) {
    if (!nbytes || !fmt || (nfiles && !files)) return PBSGPU_E_INVALID;
    uint64_t n = (fmt->with_start ? 16 : 0) + (fmt->with_tail ? 16 : 0);
    for (uint32_t i = 0; i < nfiles; ++i) n += 16 + files[i].length;
    *nbytes = n;
    return PBSGPU_OK;
}

}  // extern "C"
