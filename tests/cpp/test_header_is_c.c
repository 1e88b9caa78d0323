/* include/kmx.h is a C header: this file is compiled as C99 (tests/test_capi_cpu.py) and links nothing. */
#include <kmx.h>

int use_the_declarations(void)
{
    kmx_options o;
    kmx_index* ix = 0;
    kmx_result* r = 0;
    const uint64_t* hit_off;
    const uint32_t* positions;
    const uint8_t *status, *kinds;
    uint32_t n_parts = 0, n_dev = 0;
    int32_t devs[KMX_MAX_DEVICES];
    o.struct_size = (uint32_t)sizeof o;
    o.device = -1;
    o.n_devices = 0;
    if (kmx_version() != KMX_VERSION) return 1;
    if (kmx_index_build(0, 0, 4, 0, 0, &o, &ix) == KMX_OK) return 2;
    if (kmx_search_batch(ix, 0, 0, 0, KMX_SEARCH_DEFAULT | KMX_SEARCH_KEEP_MASKS, &r) == KMX_OK) return 3;
    (void)kmx_result_view(r, &hit_off, &positions, &status, &kinds);
    (void)kmx_result_parts(r, &n_parts);
    (void)kmx_index_devices(ix, &n_dev, devs);
    return (int)(kmx_fast_pow(4, 10) != 1048576u);
}
