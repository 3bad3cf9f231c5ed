// strip_kernels.hip -- instantiations of the strip matcher (sad_strip.inc) and their pickers.  Compiled three times,
// -DSADBM_STRIP_PART_NG=4 / 8 / 16 (64 / 128 / 256 disparities per block), so the 3 x 22 kernels build in parallel.
#include "sadbm_internal.hpp"

#ifndef SADBM_STRIP_PART_NG
#error "compile with -DSADBM_STRIP_PART_NG=4, 8 or 16"
#endif

namespace sadbm_detail {

strip_kernel_t strip_part_4(int win);
strip_kernel_t strip_part_8(int win);
strip_kernel_t strip_part_16(int win);

#include "sad_strip.inc"

namespace {
#ifndef SADBM_STRIP_WINS
#define SADBM_STRIP_WINS 25
#endif
// host copies of the kernel's compile-time geometry: staging passes per row image, dwords per row image
template <int WIN>
static void strip_geometry_one(int nsl, int *pr, int *roww)
{
    if constexpr (WIN >= 4 && WIN <= SADBM_STRIP_WINS && strip::SG<WIN>::ok) {
        using SP = strip::StagePlan<WIN>;
        *pr = ((SP::n_single(nsl) + 63) >> 6) + ((SP::n_double(nsl) + 63) >> 6);
        *roww = nsl == strip::NS + 4 ? strip::Lay<WIN, strip::NS + 4>::ROWW
              : nsl == strip::NS + 8 ? strip::Lay<WIN, strip::NS + 8>::ROWW : strip::Lay<WIN, strip::NS + 16>::ROWW;
    } else {
        (void)nsl; *pr = 0; *roww = 0;
    }
}

template <int... I>
void strip_geometry_table(int win, int nsl, int *pr, int *roww, Seq<I...>)
{
    typedef void (*fn_t)(int, int *, int *);
    static const fn_t table[] = { strip_geometry_one<I + 1>... };
    table[win - 1](nsl, pr, roww);
}

template <int WIN, int NSLM>
static strip_kernel_t strip_entry()
{
#ifdef SADBM_ONLY_WIN
    if constexpr (WIN != SADBM_ONLY_WIN) return nullptr; else                                       // experiment builds: one window
#endif
    if constexpr (WIN >= 4 && WIN <= SADBM_STRIP_WINS && strip::SG<WIN>::ok) return strip::sad_strip_kernel<WIN, NSLM>;
    else return nullptr;
}

template <int... I>
strip_kernel_t strip_part_table(int win, Seq<I...>)
{
    static const strip_kernel_t t[] = { strip_entry<I + 1, strip::NS + SADBM_STRIP_PART_NG>()... };
    return t[win - 1];
}
}  // namespace

#define SADBM_CAT2(a, b) a##b
#define SADBM_CAT(a, b) SADBM_CAT2(a, b)
// one translation unit per block size (NG = 4, 8, 16 disparity groups): strip_part_4 / _8 / _16
strip_kernel_t SADBM_CAT(strip_part_, SADBM_STRIP_PART_NG)(int win) { return strip_part_table(win, MakeSeq<MAXWIN>::type()); }

#if SADBM_STRIP_PART_NG == 4
void strip_geometry(int win, int nsl, int *pr, int *roww)
{
    if (win < 1 || win > MAXWIN) { *pr = 0; *roww = 0; return; }
    strip_geometry_table(win, nsl, pr, roww, MakeSeq<MAXWIN>::type());
}
strip_kernel_t pick_strip_kernel(int win, int nslm)
{
    if (win < 1 || win > MAXWIN) return nullptr;
    return nslm == strip::NS + 4 ? strip_part_4(win) : nslm == strip::NS + 8 ? strip_part_8(win) : strip_part_16(win);
}
#endif

}  // namespace sadbm_detail
