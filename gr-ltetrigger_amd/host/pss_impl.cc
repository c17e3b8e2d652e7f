// gr::ltetrigger::pss on the gfx950 library.
//
// Control flow is the reference's pss block (lib/pss_impl.cc:155-223): every call looks at one
// 9600-sample window that starts at the first unconsumed item; a search call asks the device for
// the lag-metric peak and its peak-to-side-lobe ratio; the ratio drives the tracking score; a
// hit (or a pending "tracking lost") emits the half-frame that starts one slot before the PSS
// peak and realigns the stream to it.  All arithmetic happens behind lt_pss_* / lt_cfo_*.
#include "pss_impl.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace gr {
namespace ltetrigger {

const pmt::pmt_t pss_impl::tracking_lost_tag_key = pmt::intern("tracking_lost");

pss::sptr pss::make(int N_id_2, float psr_threshold, int track_after, int track_every)
{
  return gnuradio::get_initial_sptr(new pss_impl(N_id_2, psr_threshold, track_after, track_every, 0));
}

pss_impl::pss_impl(int N_id_2, float psr_threshold, int track_after, int track_every, int device)
    : gr::block("pss", gr::io_signature::make(1, 1, sizeof(lt_cf32)), gr::io_signature::make(1, 1, sizeof(lt_cf32))),
      d_N_id_2(N_id_2), d_psr_threshold(psr_threshold), d_track_after(track_after), d_track_every(track_every)
{
  // the three srsLTE init calls of lib/pss_impl.cc:72-79, same failure behaviour
  if (lt_pss_init(&d_pss, device, half_frame_length, symbol_sz))
    throw std::runtime_error(std::string("Error initializing PSS: ") + lt_last_error());
  if (lt_pss_set_N_id_2(d_pss, N_id_2)) {
    lt_pss_free(d_pss);
    throw std::runtime_error("Error initializing PSS N_id_2");
  }
  if (lt_cfo_init(&d_cfo, device, half_frame_length)) {
    lt_pss_free(d_pss);
    throw std::runtime_error(std::string("Error initializing CFO: ") + lt_last_error());
  }
  set_history(half_frame_length);
  set_output_multiple(half_frame_length);
}

pss_impl::~pss_impl()
{
  lt_pss_free(d_pss);
  lt_cfo_free(d_cfo);
}

void pss_impl::forecast(int noutput_items, gr_vector_int &required)
{
  for (auto &r : required) r = noutput_items + (int)history() - 1 + max_lookahead;
}

float pss_impl::ring_mean(const float *ring, size_t count)
{
  // compute_moving_avg, lib/pss_impl.cc:94-109: first min(count, 200) entries, double accumulator
  if (!count) return 0.0f;
  if (count > (size_t)ring_size) count = ring_size;
  double acc = 0.0;
  for (size_t i = 0; i < count; ++i) acc += ring[i];
  return (float)(acc / (double)count);
}

void pss_impl::score_hit() // incr_score, lib/pss_impl.cc:111-127
{
  if (d_is_tracking && d_score == d_track_after) return;
  ++d_score;
  if (!d_is_tracking && d_score == d_track_after) {
    d_is_tracking = true;
    lt_pss_reset(d_pss); // start tracking on a clean lag-metric average
  }
}

void pss_impl::score_miss() // reset_score, lib/pss_impl.cc:129-152
{
  if (d_score == 0) return; // nothing to lose: keep accumulating the average
  d_score = 0;
  d_timer = 0; // search again on the very next half-frame
  d_is_tracking = false;
  lt_pss_reset(d_pss);
  {
    std::lock_guard<std::mutex> l(d_ring_mu);
    d_psr_count = 0;
    d_cfo_count = 0;
  }
  lt_cfo_set_last_freq(d_cfo, 0.f);
  d_tracking_lost = true; // tag the next emitted half-frame
}

int pss_impl::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items)
{
  const lt_cf32 *in = static_cast<const lt_cf32 *>(input_items[0]) + (history() - 1);
  lt_cf32 *out = static_cast<lt_cf32 *>(output_items[0]);
  const int fresh = ninput_items[0] - (int)(history() - 1); // items at and after `in`

  if (!d_is_tracking || d_timer == 0) {
    d_timer = d_track_every;
    const int pos = lt_pss_find_pss(d_pss, in, &d_psr);
    if (pos < 0) throw std::runtime_error(std::string("lt_pss_find_pss: ") + lt_last_error());
    d_peak_pos = pos;
    std::lock_guard<std::mutex> l(d_ring_mu);
    d_psr_ring[d_psr_count++ % ring_size] = d_psr;
  } else {
    --d_timer;
  }

  const bool hit = d_psr > psr_threshold();
  if (hit) score_hit();
  else score_miss();
  if (d_psr > max_psr()) d_psr_max.store(d_psr, std::memory_order_relaxed);

  if (!hit && !d_tracking_lost) { // nothing to see: drop this half-frame
    consume_each(half_frame_length);
    return 0;
  }

  // the half-frame begins one slot before the end of the PSS symbol; realign the stream to it
  const int frame_start = d_peak_pos - slot_length; // may be negative: reaches into the history
  d_peak_pos = slot_length;
  const int nconsume = frame_start + half_frame_length;
  if (nconsume > fresh) {
    // only possible while draining the tail of a finite capture (forecast guarantees the
    // look-ahead otherwise): the reference would read past its input here (quirk q1)
    return WORK_DONE;
  }
  std::memcpy(out, in + frame_start, sizeof(lt_cf32) * half_frame_length);
  consume_each(nconsume);

  if (d_is_tracking) {
    float cfo = 0.f;
    if (lt_pss_cfo_compute(d_pss, out + (slot_length - symbol_sz), &cfo))
      throw std::runtime_error(std::string("lt_pss_cfo_compute: ") + lt_last_error());
    {
      std::lock_guard<std::mutex> l(d_ring_mu);
      d_cfo_ring[d_cfo_count++ % ring_size] = cfo;
    }
    if (lt_cfo_correct(d_cfo, out, out, -mean_cfo() / symbol_sz))
      throw std::runtime_error(std::string("lt_cfo_correct: ") + lt_last_error());
    // srslte_pss_chest (lib/pss_impl.cc:206-209) filled a buffer nobody reads: not reproduced
  } else {
    add_item_tag(0, nitems_written(0), tracking_lost_tag_key, PMT_NIL);
    d_tracking_lost = false;
  }
  (void)noutput_items;
  return half_frame_length;
}

} // namespace ltetrigger
} // namespace gr
