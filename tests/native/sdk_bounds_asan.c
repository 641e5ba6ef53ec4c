/* CPU, AddressSanitizer: the host side of the SDK decode calls -- picture length from the section table
 * (hvq_picture_length) and the host entropy parse -- on EXACTLY sized heap buffers (no slack after the last byte), for legal
 * pictures, truncated ones and bit-flipped ones.  Any read past the allocation aborts the process.
 * usage: sdk_bounds_asan <w> <h> <is15> <file with records: u32 frame_type, u32 len, len bytes ...> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/hvqm4_amd.h"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int w = atoi(argv[1]), h = atoi(argv[2]), is15 = atoi(argv[3]);
    FILE *f = fopen(argv[4], "rb");
    if (!f) return 2;
    HvqParser *p = hvq_parser_create(w, h, 2, 2, is15);
    if (!p) return 3;
    const size_t cap = hvq_parser_blob_bound(p);
    uint8_t *blob = malloc(cap);
    uint32_t hd[2];
    unsigned n = 0, parsed = 0, refused = 0, exact = 0;
    uint32_t seed = 12345;
    while (fread(hd, 4, 2, f) == 2) {
        uint8_t *src = malloc(hd[1]);
        if (fread(src, 1, hd[1], f) != hd[1]) return 4;
        for (int variant = 0; variant < 12; ++variant) {
            size_t len = hd[1];
            if (variant >= 1 && variant <= 3) len = len * (size_t)variant / 4 + 0x50;      /* truncated */
            if (len > hd[1]) len = hd[1];
            uint8_t *pic = malloc(len);                                   /* exactly sized: ASan guards the next byte */
            memcpy(pic, src, len);
            if (variant >= 4)
                for (int k = 0; k < 6 * (variant - 3); ++k) { seed = seed * 1664525u + 1013904223u; pic[(seed >> 8) % len] ^= (uint8_t)(1u << (seed & 7)); }
            size_t got = 0, blen = 0;
            const int rc = hvq_picture_length(pic, (int)hd[0], (uint32_t)len, &got);
            if (rc == 0) {
                if (got > len) { fprintf(stderr, "length %zu beyond the %zu-byte frame\n", got, len); return 5; }
                if (variant == 0) { if (got != len) { fprintf(stderr, "legal picture: length %zu, expected %zu\n", got, len); return 6; } ++exact; }
                if (hvq_parse_picture(p, (int)hd[0], pic, got, blob, cap, &blen) == 0) ++parsed; else ++refused;
            } else ++refused;
            free(pic);
            ++n;
        }
        free(src);
    }
    printf("%u pictures x variants: %u parsed, %u refused, %u legal lengths exact\n", n, parsed, refused, exact);
    hvq_parser_destroy(p);
    free(blob);
    return exact ? 0 : 7;
}
