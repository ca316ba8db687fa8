/* A plain-C client of include/mi355vits.h: proves the header is C (not C++), that the structs have the layout the
 * Python side assumes, and walks the whole call sequence of INTEGRATION.md §5 against whatever libmi355vits*.so it is
 * linked with.  usage: abi_client <voice.m355> <out.raw>   (writes int16 PCM of a fixed 5-phoneme utterance) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355vits.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    printf("version %s\n", mi355vits_version());
    printf("sizeof config %zu run_args %zu result %zu\n", sizeof(mi355vits_config), sizeof(mi355vits_run_args),
           sizeof(mi355vits_result));
    mi355vits_handle h = NULL;
    /* error path first: a file that is not a container */
    int rc = mi355vits_create(argv[0], 0, &h);
    if (rc == MI355VITS_OK || h != NULL) { fprintf(stderr, "loading a non-container succeeded\n"); return 1; }
    printf("expected failure rc=%d msg=%s\n", rc, mi355vits_last_error(NULL));
    rc = mi355vits_create(argv[1], 0, &h);
    if (rc != MI355VITS_OK) { fprintf(stderr, "create: %d %s\n", rc, mi355vits_last_error(NULL)); return 1; }
    mi355vits_config cfg;
    if (mi355vits_get_config(h, &cfg) != MI355VITS_OK) return 1;
    printf("hidden %d symbols %d speakers %d\n", cfg.hidden_channels, cfg.num_symbols, cfg.n_speakers);
    int64_t ids[5] = {3, 7, 1, 9, 4};
    int64_t lengths[1] = {5};
    int64_t sid[1] = {0};
    float scales[3] = {0.0f, 1.0f, 0.0f};
    mi355vits_run_args a;
    memset(&a, 0, sizeof a);
    a.batch = 1; a.tx_max = 5; a.ids = ids; a.lengths = lengths; a.scales = scales;
    a.sid = cfg.n_speakers > 1 ? sid : NULL;
    a.flags = MI355VITS_WANT_FLOAT | MI355VITS_WANT_PCM16;
    mi355vits_result r;
    memset(&r, 0, sizeof r);
    rc = mi355vits_run(h, &a, &r);
    if (rc != MI355VITS_OK) { fprintf(stderr, "run: %d %s\n", rc, mi355vits_last_error(h)); return 1; }
    printf("samples %lld peak %.6f\n", (long long)r.lengths[0], r.peaks[0]);
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(r.pcm, sizeof(int16_t), (size_t)r.lengths[0], f);
    fclose(f);
    mi355vits_free_result(&r);
    /* bad argument: id out of range -> error code, handle stays usable */
    ids[2] = 100000;
    rc = mi355vits_run(h, &a, &r);
    if (rc != MI355VITS_ERR_INVALID) { fprintf(stderr, "out-of-range id gave rc=%d\n", rc); return 1; }
    printf("expected failure rc=%d msg=%s\n", rc, mi355vits_last_error(h));
    mi355vits_destroy(h);
    return 0;
}
