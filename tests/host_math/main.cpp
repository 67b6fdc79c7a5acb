/* TEST INFRASTRUCTURE ONLY - stand-alone driver of harness.cpp for the sanitizer build (an instrumented shared object cannot be
 * loaded into an uninstrumented python): reads faces [nf, 9] float32 from a file, prints the counters of hm_compare. */
#include <stdio.h>
#include <stdlib.h>
extern "C" void hm_compare(const float* faces, long nf, int IS, float sigma, float dist_eps_log, int max_px, long* counts);
int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s faces.bin IS sigma dist_eps_log max_px\n", argv[0]); return 2; }
    FILE* fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    fseek(fh, 0, SEEK_END);
    const long bytes = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    float* faces = (float*)malloc(bytes ? bytes : 4);
    if (fread(faces, 1, bytes, fh) != (size_t)bytes) return 2;
    fclose(fh);
    long counts[24] = {0};
    hm_compare(faces, bytes / 36, atoi(argv[2]), (float)atof(argv[3]), (float)atof(argv[4]), atoi(argv[5]), counts);
    for (int i = 0; i < 18; i++) printf("%ld%c", counts[i], i == 17 ? '\n' : ' ');
    free(faces);
    return 0;
}
