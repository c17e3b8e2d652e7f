/* C interface to the host blocks + mini runtime, for the Python layer (ctypes).
 * This is NOT the drop-in boundary (that is include/ltetrigger_hip.h); it plays the role SWIG
 * plays in the reference (swig/ltetrigger_swig.i): constructing blocks, wiring a flowgraph,
 * running it and reading results back from Python. */
#ifndef LTETRIGGER_HOST_CAPI_H
#define LTETRIGGER_HOST_CAPI_H
#include <stdint.h>

#include "../../include/ltetrigger_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ltr_tb ltr_tb;

ltr_tb *ltr_tb_create(void);
void ltr_tb_free(ltr_tb *tb);
const char *ltr_last_error(void);

/* every ltr_add_* returns a block id >= 0, or -1 */
int ltr_add_vector_source(ltr_tb *tb, const lt_cf32 *data, int64_t n, int repeat);
int ltr_add_file_source(ltr_tb *tb, const char *path, int repeat);
int ltr_add_head(ltr_tb *tb, int64_t nitems);
int ltr_add_resampler(ltr_tb *tb, int decimation);
int ltr_add_pss(ltr_tb *tb, int N_id_2, float psr_threshold, int track_after, int track_every);
int ltr_add_sss(ltr_tb *tb, int N_id_2);
int ltr_add_mib(ltr_tb *tb, int exit_on_success);
int ltr_add_cellstore(ltr_tb *tb);
int ltr_add_message_debug(ltr_tb *tb);
int ltr_add_null_sink(ltr_tb *tb);
int ltr_add_probe(ltr_tb *tb); /* sink that keeps every item and tag it receives */

int ltr_connect(ltr_tb *tb, int src, int dst);
int ltr_msg_connect(ltr_tb *tb, int src, const char *src_port, int dst, const char *dst_port);
int ltr_run(ltr_tb *tb);
int ltr_run_timeout(ltr_tb *tb, double max_seconds); /* < 0: no limit */
/* the same flowgraph under a thread-per-block scheduler (GNU Radio's default): callable while other threads poll the
 * ltr_pss_* getters, ltr_msgdebug_count and the ltr_cellstore_* calls of the same top block */
int ltr_run_threaded(ltr_tb *tb, double max_seconds);

float ltr_pss_max_psr(ltr_tb *tb, int id);
float ltr_pss_mean_psr(ltr_tb *tb, int id);
float ltr_pss_mean_cfo(ltr_tb *tb, int id);
float ltr_pss_psr_threshold(ltr_tb *tb, int id);
float ltr_pss_tracking_score(ltr_tb *tb, int id);
int ltr_pss_set_psr_threshold(ltr_tb *tb, int id, float t);

int ltr_msgdebug_count(ltr_tb *tb, int id);
int ltr_msgdebug_json(ltr_tb *tb, int id, int idx, char *buf, int len);

int ltr_cellstore_tracking(ltr_tb *tb, int id);
int ltr_cellstore_cells_json(ltr_tb *tb, int id, char *buf, int len);  /* JSON array */
int ltr_cellstore_latest_json(ltr_tb *tb, int id, char *buf, int len); /* dict or null */

int64_t ltr_probe_nitems(ltr_tb *tb, int id);
int64_t ltr_probe_items(ltr_tb *tb, int id, lt_cf32 *out, int64_t cap);
int ltr_probe_tags_json(ltr_tb *tb, int id, char *buf, int len); /* [{"offset":n,"key":"k","value":v},...] */

/* stand-alone PBCH/MIB decode of one 1920-sample subframe (used by the batched Python path) */
typedef struct {
  int32_t nof_prb, nof_ports, phich_length_ext, phich_resources, sfn, frame_idx;
} ltr_mib;
int ltr_decode_mib(const lt_cf32 *subframe1920, int cell_id, int cp_ext, int nof_ports_hint, ltr_mib *out);

#ifdef __cplusplus
}
#endif
#endif
