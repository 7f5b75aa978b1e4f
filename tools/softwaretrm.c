/* softwaretrm.c -- the `softwareTRM` command line tool (Frameworks/Tube/main.m:12-67) over libtrm_hip.so's C ABI:
 *
 *     softwaretrm [-v] inputFile outputFile
 *
 * parses a .trm / Monet.parameters file (TRMDataList -initWithContentsOfFile:), builds the tube, synthesizes on the
 * GPU and writes the AU / AIFF / WAVE file the input names.  Messages and exit codes are main.m's: usage and
 * "Aborting..." on stderr with exit(-1); a failed save is reported and -- like main.m:57-62 -- does not change the
 * exit code.  Plain C: the caller side of include/trm_c_api.h that INTEGRATION.md section 3 describes.
 *
 * Batch directory mode (no reference counterpart: the reference runs one file per process):
 *
 *     softwaretrm --batch inputDir outputDir
 *
 * synthesizes every *.trm / *.parameters file of inputDir; files that share their utterance-rate parameters go to the
 * GPU as ONE launch (trm_batch_synthesize_host), and each gets its own sound file (same name, the extension of its
 * outputFileFormat), byte for byte what the single-file mode writes.
 */
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "../include/trm_c_api.h"

static int ends_with(const char *s, const char *suffix)
{
    size_t n = strlen(s), m = strlen(suffix);
    return n >= m && !strcmp(s + n - m, suffix);
}

static int by_name(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

typedef struct {
    char *name;
    trm_input_params params;
    trm_parameters *frames;
    size_t nframes;
    int done;
} utterance;

static int batch_mode(const char *indir, const char *outdir)
{
    static const char *const ext[3] = {".au", ".aiff", ".wav"};
    DIR *d = opendir(indir);
    if (!d) { fprintf(stderr, "%s: cannot open directory\n", indir); return 255; }
    char **names = NULL;
    size_t n = 0, cap = 0;
    for (struct dirent *e; (e = readdir(d));) {
        if (!ends_with(e->d_name, ".trm") && !ends_with(e->d_name, ".parameters")) continue;
        if (n == cap) { cap = cap ? 2 * cap : 64; names = (char **)realloc(names, cap * sizeof *names); }
        names[n++] = strdup(e->d_name);
    }
    closedir(d);
    qsort(names, n, sizeof *names, by_name);
    mkdir(outdir, 0777);

    utterance *u = (utterance *)calloc(n ? n : 1, sizeof *u);
    char path[4096];
    size_t nu = 0;
    for (size_t i = 0; i < n; i++) {
        snprintf(path, sizeof path, "%s/%s", indir, names[i]);
        memset(&u[nu].params, 0, sizeof u[nu].params);      /* (padding bytes: the params are compared with memcmp) */
        if (trm_data_list_read_file(path, &u[nu].params, &u[nu].frames, &u[nu].nframes)) {
            fprintf(stderr, "%s: cannot parse, skipped\n", names[i]);
            continue;
        }
        u[nu++].name = names[i];
    }

    size_t nfiles = 0, nlaunches = 0;
    for (size_t i = 0; i < nu; i++) {
        if (u[i].done) continue;
        /* the group: every utterance with these utterance-rate parameters */
        size_t V = 0, rows = 0, samples = 0;
        trm_batch *b = NULL;
        if (trm_batch_create(&u[i].params, -1, &b)) { fprintf(stderr, "%s: %s\n", u[i].name, trm_last_error()); return 1; }
        for (size_t j = i; j < nu; j++)
            if (!u[j].done && !memcmp(&u[j].params, &u[i].params, sizeof u[i].params)) {
                V++;
                rows += u[j].nframes;
                samples += trm_batch_samples_for_frames(b, u[j].nframes);
            }
        float *frames = (float *)malloc((rows ? rows : 1) * 16 * sizeof(float));
        float *pcm = (float *)malloc((samples ? samples : 1) * sizeof(float));
        uint64_t *foff = (uint64_t *)malloc(V * sizeof *foff), *ooff = (uint64_t *)malloc(V * sizeof *ooff);
        uint32_t *nfr = (uint32_t *)malloc(V * sizeof *nfr), *ns = (uint32_t *)malloc(V * sizeof *ns);
        float *mx = (float *)malloc(V * sizeof *mx);
        size_t *member = (size_t *)malloc(V * sizeof *member);
        size_t v = 0, r = 0, o = 0;
        for (size_t j = i; j < nu; j++) {
            if (u[j].done || memcmp(&u[j].params, &u[i].params, sizeof u[i].params)) continue;
            member[v] = j; foff[v] = r; ooff[v] = o; nfr[v] = (uint32_t)u[j].nframes;
            const double *src = (const double *)u[j].frames;          /* trm_parameters = 16 doubles in column order */
            for (size_t k = 0; k < u[j].nframes * 16; k++) frames[r * 16 + k] = (float)src[k];
            r += u[j].nframes;
            o += trm_batch_samples_for_frames(b, u[j].nframes);
            v++;
        }
        if (trm_batch_synthesize_host(b, V, frames, foff, nfr, pcm, ooff, ns, mx)) {
            fprintf(stderr, "%s: %s\n", u[i].name, trm_last_error());
            return 1;
        }
        nlaunches++;
        for (v = 0; v < V; v++) {
            utterance *m = &u[member[v]];
            char stem[1024];
            snprintf(stem, sizeof stem, "%s", m->name);
            char *dot = strrchr(stem, '.');
            if (dot) *dot = 0;
            int f = m->params.outputFileFormat;
            snprintf(path, sizeof path, "%s/%s%s", outdir, stem, ext[f >= 0 && f < 3 ? f : 0]);
            if (trm_write_sound_file(&m->params, pcm + ooff[v], ns[v], mx[v], path))
                fprintf(stderr, "Failed to save output: %s\n", trm_last_error());
            else
                nfiles++;
            m->done = 1;
        }
        trm_batch_destroy(b);
        free(frames); free(pcm); free(foff); free(ooff); free(nfr); free(ns); free(mx); free(member);
    }
    printf("%zu files in %zu launches\n", nfiles, nlaunches);
    for (size_t i = 0; i < nu; i++) trm_free(u[i].frames);
    for (size_t i = 0; i < n; i++) free(names[i]);
    free(names); free(u);
    return 0;
}

int main(int argc, char *argv[])
{
    int verbose = 0;
    const char *inputFile, *outputFile;

    if (argc == 4 && !strcmp("--batch", argv[1])) return batch_mode(argv[2], argv[3]);
    if (argc == 3) {                                                /* main.m:18-29 */
        inputFile = argv[1];
        outputFile = argv[2];
    } else if (argc == 4 && !strcmp("-v", argv[1])) {
        verbose = 1;
        inputFile = argv[2];
        outputFile = argv[3];
    } else {
        fprintf(stderr, "Usage:  %s [-v] inputFile outputFile\n", argv[0]);
        exit(-1);
    }

    trm_input_params params;                                        /* main.m:31-35 */
    trm_parameters *frames = NULL;
    size_t nframes = 0;
    if (trm_data_list_read_file(inputFile, &params, &frames, &nframes)) {
        fprintf(stderr, "Aborting...\n");
        exit(-1);
    }

    trm_tube *tube = NULL;                                          /* main.m:38-42 */
    if (trm_tube_create(&params, -1, &tube)) {
        fprintf(stderr, "Aborting...\n");
        exit(-1);
    }

    if (verbose) {                                                  /* main.m:44-51 */
        printf("input file:\t\t%s\n\n", inputFile);
        trm_tube_print_input_data(tube, frames, nframes);
        printf("\nCalculating floating point samples...");
        printf("\nStarting synthesis\n");
        fflush(stdout);
    }

    if (trm_tube_synthesize(tube, frames, nframes)) {               /* main.m:53; -synthesize returns void: the reference cannot fail here */
        fprintf(stderr, "%s\nAborting...\n", trm_last_error());
        exit(-1);
    }

    if (verbose) printf("done.\n");

    if (trm_tube_save_output_to_file(tube, outputFile))             /* main.m:58-61 */
        fprintf(stderr, "Failed to save output: %s\n", trm_last_error());

    if (verbose) printf("\nWrote scaled samples to file:  %s\n", outputFile);

    trm_tube_destroy(tube);
    trm_free(frames);
    return 0;
}
