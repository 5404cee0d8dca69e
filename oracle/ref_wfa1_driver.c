/* ORACLE tooling (test infrastructure): driver around WFA v1, the C library in
 * the reference's cargo cache (libwfa-0.1.2.crate, MIT).  WFA v1 is NOT what
 * seqrush links (it links WFA2-lib through lib_wfa2); it is single-piece
 * gap-affine with full memory, and is used only as a secondary score oracle
 * and to reproduce the reference's WFA boundary known answers
 * (tests/test_wfa2_cigar_debug.rs, tests/test_cigar_validity.rs).
 * stdin : lines "pattern text mismatch gap_open gap_ext"
 * stdout: lines "score cigar"   (score >= 0 = penalty, raw alphabet M X I D)
 * Built by oracle/build_ref.sh into oracle/_ref/ ; sources stay where they lie. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gap_affine/affine_wavefront_align.h"

int main(void) {
    static char p[1 << 16], t[1 << 16];
    int x, o, e;
    while (scanf("%65535s %65535s %d %d %d", p, t, &x, &o, &e) == 5) {
        const int plen = (int)strlen(p), tlen = (int)strlen(t);
        mm_allocator_t *const mm = mm_allocator_new(BUFFER_SIZE_8M);
        affine_penalties_t pen = {.match = 0, .mismatch = x, .gap_opening = o, .gap_extension = e};
        affine_wavefronts_t *aw = affine_wavefronts_new_complete(plen, tlen, &pen, NULL, mm);
        affine_wavefronts_align(aw, p, plen, t, tlen);
        const int score = edit_cigar_score_gap_affine(&aw->edit_cigar, &pen);
        printf("%d ", -score);
        for (int i = aw->edit_cigar.begin_offset; i < aw->edit_cigar.end_offset; i++)
            putchar(aw->edit_cigar.operations[i]);
        putchar('\n');
        affine_wavefronts_delete(aw);
        mm_allocator_delete(mm);
    }
    return 0;
}
