/* Native test driver for the host-only part of libpbsgpu (hostonly.cpp + reuse.cpp), built with
 * -fsanitize=address,undefined by tests/test_native_sanitized.py. The reference leans on
 * `go test -race` (.github/workflows/go-tests.yml:24-33); this is the counterpart for the new native
 * host code: out-of-bounds, misaligned and overflowing accesses in the index arithmetic abort the run. */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pbsgpu.h"

static void make_idx(pbsgpu_record *idx, int n, uint64_t size) {
    memset(idx, 0, sizeof(*idx) * (size_t)n);
    for (int i = 0; i < n; i++) {
        idx[i].end = (uint64_t)(i + 1) * size;
        idx[i].size = (uint32_t)size;
        idx[i].digest[0] = (uint8_t)i;
        idx[i].digest[1] = (uint8_t)(i >> 8);
    }
}

int main(void) {
    /* config */
    pbsgpu_config c;
    assert(pbsgpu_config_init(4u << 20, NULL, &c) == PBSGPU_OK);
    assert(c.min == (1u << 20) && c.max == (16u << 20) && c.mask == 0x7FFFFF && c.break_min == 0x7FFFFD && c.window == 64);
    assert(pbsgpu_config_init(3000, NULL, &c) == PBSGPU_E_INVALID);
    assert(pbsgpu_config_init(128, NULL, &c) == PBSGPU_E_INVALID);
    assert(pbsgpu_config_init(4096, NULL, NULL) == PBSGPU_E_INVALID);
    assert(memcmp(c.table, pbsgpu_default_table(), sizeof(c.table)) == 0);
    assert(strcmp(pbsgpu_strerror(PBSGPU_OK), "ok") == 0 && pbsgpu_strerror(-12345) != NULL);

    /* reuse planner: the reference table (commit_bottleneck_test.go:795-835) on exact-size heap buffers */
    struct { uint64_t rs, re, n, sp, ep; } cases[] = {
        {0, 500, 5, 0, 0}, {0, 100, 2, 0, 100}, {400, 500, 1, 0, 0}, {100, 300, 3, 0, 100}, {50, 300, 4, 50, 100},
        {100, 350, 3, 0, 50}, {50, 350, 4, 50, 50}, {10, 20, 1, 10, 80}, {100, 100, 0, 0, 0}, {600, 700, 0, 0, 0}};
    pbsgpu_record *idx = (pbsgpu_record *)malloc(5 * sizeof(pbsgpu_record));
    make_idx(idx, 5, 100);
    for (unsigned k = 0; k < sizeof(cases) / sizeof(cases[0]); k++) {
        uint64_t n = 0, sp = 0, ep = 0;
        assert(pbsgpu_reuse_lookup(idx, 5, cases[k].rs, cases[k].re, NULL, 0, &n, &sp, &ep) == PBSGPU_OK);
        assert(n == cases[k].n && sp == cases[k].sp && ep == cases[k].ep);
        pbsgpu_reuse_chunk *out = (pbsgpu_reuse_chunk *)malloc((n ? n : 1) * sizeof(*out)); /* exact size: ASAN guards the end */
        assert(pbsgpu_reuse_lookup(idx, 5, cases[k].rs, cases[k].re, out, n, &n, &sp, &ep) == PBSGPU_OK);
        if (n > 1) { /* too small a buffer must be reported, not overrun */
            uint64_t n2 = 0;
            assert(pbsgpu_reuse_lookup(idx, 5, cases[k].rs, cases[k].re, out, n - 1, &n2, &sp, &ep) == PBSGPU_E_CAPACITY);
            assert(n2 == n);
        }
        free(out);
    }
    free(idx);
    idx = (pbsgpu_record *)malloc(10 * sizeof(pbsgpu_record));
    make_idx(idx, 10, 1000);
    int reuse = -1;
    assert(pbsgpu_reuse_should(idx, 10, 0, 916, NULL, 0.1, &reuse) == PBSGPU_OK && reuse == 1);
    assert(pbsgpu_reuse_should(idx, 10, 500, 526, NULL, 0.1, &reuse) == PBSGPU_OK && reuse == 0);
    assert(pbsgpu_reuse_should(NULL, 0, 0, 100, NULL, 0.1, &reuse) == PBSGPU_OK && reuse == 1);
    assert(pbsgpu_reuse_should(idx, 10, ~0ull - 5, ~0ull, NULL, 0.1, &reuse) == PBSGPU_OK && reuse == 1); /* past the end */

    /* dynamic index decode: exact-size buffers, truncated and corrupt inputs */
    uint64_t nb = 0;
    assert(pbsgpu_didx_size(3, &nb) == PBSGPU_OK && nb == 4096 + 120);
    uint8_t *img = (uint8_t *)calloc(1, (size_t)nb);
    const uint8_t magic[8] = {28, 145, 78, 165, 25, 186, 179, 205};
    memcpy(img, magic, 8);
    for (int i = 0; i < 3; i++) {
        uint64_t end = (uint64_t)(i + 1) * 1000;
        memcpy(img + 4096 + 40 * i, &end, 8);
        memset(img + 4096 + 40 * i + 8, i + 1, 32);
    }
    pbsgpu_record out3[3];
    uint64_t n = 0;
    int64_t ctime = -1;
    uint8_t csum[32];
    assert(pbsgpu_didx_decode(img, nb, out3, 3, &n, &ctime, csum) == PBSGPU_OK && n == 3 && ctime == 0);
    assert(out3[2].end == 3000 && out3[2].size == 1000 && out3[2].digest[31] == 3);
    assert(pbsgpu_didx_decode(img, nb, out3, 2, &n, NULL, NULL) == PBSGPU_E_CAPACITY && n == 3);
    assert(pbsgpu_didx_decode(img, nb - 1, out3, 3, &n, NULL, NULL) == PBSGPU_E_INVALID);
    assert(pbsgpu_didx_decode(img, 100, out3, 3, &n, NULL, NULL) == PBSGPU_E_INVALID);
    img[0] ^= 1;
    assert(pbsgpu_didx_decode(img, nb, out3, 3, &n, NULL, NULL) == PBSGPU_E_INVALID);
    img[0] ^= 1;
    uint64_t bad = 500; /* non-monotonic end */
    memcpy(img + 4096 + 40, &bad, 8);
    assert(pbsgpu_didx_decode(img, nb, out3, 3, &n, NULL, NULL) == PBSGPU_E_INVALID);
    free(img);
    free(idx);

    /* payload sizing */
    pbsgpu_payload_format f;
    assert(pbsgpu_payload_format_default(&f) == PBSGPU_OK && f.with_start == 1 && f.with_tail == 1);
    pbsgpu_segment files[3] = {{0, 0}, {10, 100}, {200, 16}};
    assert(pbsgpu_payload_size(files, 3, &f, &nb) == PBSGPU_OK && nb == 32 + 3 * 16 + 116);
    f.with_start = f.with_tail = 0;
    assert(pbsgpu_payload_size(files, 3, &f, &nb) == PBSGPU_OK && nb == 3 * 16 + 116);
    assert(pbsgpu_payload_size(NULL, 3, &f, &nb) == PBSGPU_E_INVALID);
    puts("native-host-ok");
    return 0;
}
