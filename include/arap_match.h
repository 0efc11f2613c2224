/* arap_match.h -- C ABI of libarapmatch.so: dense matching of a frame pair on an MI355X (gfx950), the stage in front of
 * the ARAP solve.
 *
 * What it replaces: the reference shells out to an external binary, DeepMatching 1.2.2,
 *     ./deepmatching img1 img2 -nt 0 -out FILE -ngh_rad 100              (/root/reference/para_gen.py:227-240)
 * and parses its lines `x1 y1 x2 y2 score index` (/root/reference/para_gen.py:468-479).  The binary's source is not in the
 * reference tree (deepmatching/get_deepmatching.sh:3 downloads it), so this library implements the PUBLISHED algorithm
 * (Revaud, Weinzaepfel, Harchaoui, Schmid, IJCV 2016; restated section by section in oracle/dm_oracle.py, against which
 * the kernels are tested) and its parity with the binary is unpinned.  `para_gen.py --dm_bin builtin` selects it.
 *
 * No CPU fallback: ArapMatch_Create returns NULL without a HIP device.  Plain pointers and sizes only.
 */
#ifndef ARAP_MATCH_H
#define ARAP_MATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ArapMatch ArapMatch;

/* Matcher for W x H frames (both frames of a pair have this size), displacements up to ngh_rad pixels (-ngh_rad).
 * Works at half resolution like the binary's default (-downscale 1).  NULL: no GPU, or a frame smaller than 16 x 16. */
ArapMatch* ArapMatch_Create(unsigned W, unsigned H, unsigned ngh_rad);
void ArapMatch_Free(ArapMatch* m);

/* One pair: rgb1, rgb2 = HOST uint8 [H][W][3].  Writes up to `cap` matches as rows of six floats
 *     x1 y1 x2 y2 score index        (full-resolution pixel coordinates, integers; index = row number)
 * into `out` (HOST), in the order of the atomic patches of frame 1 (row major).  Returns the number of matches
 * (may exceed cap: then only cap rows were written), or -1 on bad arguments. */
int ArapMatch_Run(ArapMatch* m, const uint8_t* rgb1, const uint8_t* rgb2, float* out, unsigned cap);

/* --- introspection for the parity tests (oracle/dm_oracle.py) ------------------------------------------------------- */
/* pyramid levels of the last Run (level 0 = 4x4 atomic patches) */
int ArapMatch_Levels(ArapMatch* m);
/* level geometry: patch grid nh x nw, map size S x S, centre cell c (displacement 0) */
int ArapMatch_LevelInfo(ArapMatch* m, int level, int* nh, int* nw, int* S, int* c);
/* copy a level's maps, float [nh][nw][S][S], to HOST memory */
int ArapMatch_GetLevel(ArapMatch* m, int level, float* host);
/* copy the pixel descriptors of frame `which` (0 / 1), float [H/2][W/2][9], to HOST memory */
int ArapMatch_GetDescriptors(ArapMatch* m, int which, float* host);
/* device time of the last Run in milliseconds (HIP events on the matcher's stream) */
float ArapMatch_LastRunMs(ArapMatch* m);
/* ... and of its dominant kernel, the bottom-level correlation (k_corr0), alone */
float ArapMatch_LastCorrMs(ArapMatch* m);

#ifdef __cplusplus
}
#endif
#endif
