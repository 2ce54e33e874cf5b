/* The CPU oracle under AddressSanitizer/UBSan: exact-size heap buffers, every padding length, ragged and empty
 * segments, the streaming/whole-buffer invariant. Test infrastructure testing test infrastructure. */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/oracle.h"

int main(void) {
    oracle_config cfg;
    assert(oracle_config_init(4096, NULL, &cfg) == 0 && cfg.min == 1024 && cfg.max == 16384);
    assert(oracle_config_init(1000, NULL, &cfg) != 0);
    assert(oracle_config_init(4096, NULL, &cfg) == 0);

    const size_t n = 300007;
    uint8_t *data = (uint8_t *)malloc(n);
    oracle_fill(data, 0, n, 99, 3);
    /* whole-buffer cut list */
    uint64_t *ends = (uint64_t *)malloc(sizeof(uint64_t) * 512);
    size_t ne = oracle_chunk_stream(&cfg, data, n, ends, 512);
    assert(ne > 20 && ne <= 512 && ends[ne - 1] == n);
    /* streaming with odd feed sizes reproduces it */
    oracle_chunker c;
    oracle_chunker_init(&c, &cfg);
    size_t pos = 0, k = 0;
    while (pos < n) {
        size_t step = 1 + (pos * 7919u) % 977u;
        if (step > n - pos) step = n - pos;
        size_t off = 0;
        while (off < step) {
            size_t r = oracle_chunker_scan(&c, data + pos + off, step - off);
            if (!r) break;
            off += r;
            assert(k < ne && ends[k] == pos + off);
            k++;
        }
        pos += step;
    }
    assert(k == ne || (k == ne - 1 && ends[ne - 1] == n));
    /* candidates: exact-size output buffer */
    size_t nc = oracle_candidates(&cfg, data, n, NULL, 0);
    uint64_t *cand = (uint64_t *)malloc(sizeof(uint64_t) * (nc ? nc : 1));
    assert(oracle_candidates(&cfg, data, n, cand, nc) == nc);
    assert(oracle_candidates(&cfg, data, 63, cand, nc) == 0);
    /* sha256: every length 0..200 on exact-size copies, scalar == SHA-NI */
    for (size_t len = 0; len <= 200; len++) {
        uint8_t *p = (uint8_t *)malloc(len ? len : 1);
        memcpy(p, data + 11, len);
        uint8_t d0[32], d1[32];
        oracle_sha256(p, len, d0, 0);
        oracle_sha256(p, len, d1, 1);
        assert(memcmp(d0, d1, 32) == 0);
        free(p);
    }
    /* ragged + empty segments, exact-size record buffer */
    oracle_segment segs[5] = {{0, 100000}, {100000, 0}, {100001, 17}, {100100, 65}, {150000, n - 150000}};
    size_t nr = oracle_chunk_and_digest(&cfg, data, segs, 5, NULL, 0, 1);
    oracle_record *recs = (oracle_record *)malloc(sizeof(oracle_record) * nr);
    assert(oracle_chunk_and_digest(&cfg, data, segs, 5, recs, nr, 1) == nr);
    uint64_t sum = 0;
    for (size_t i = 0; i < nr; i++) sum += recs[i].size;
    assert(sum == 100000 + 0 + 17 + 65 + (n - 150000));
    /* fill at unaligned offsets/lengths */
    uint8_t *f = (uint8_t *)malloc(1001);
    for (unsigned kind = 0; kind < 4; kind++) oracle_fill(f, 8 * 13, 1001, 5, kind);
    free(f); free(recs); free(cand); free(ends); free(data);
    puts("oracle-asan-ok");
    return 0;
}
