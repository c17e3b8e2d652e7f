// Private implementation of gr::ltetrigger::pss on top of the gfx950 C-ABI.
// State and constants follow the reference's lib/pss_impl.h:31-81; the srslte_pss_t / srslte_cfo_t
// members become opaque lt_pss / lt_cfo handles.
#pragma once
#include <atomic>
#include <cstddef>
#include <mutex>

#include "../../include/ltetrigger_hip.h"
#include "ltetrigger/pss.h"

namespace gr {
namespace ltetrigger {

class pss_impl : public pss
{
public:
  pss_impl(int N_id_2, float psr_threshold, int track_after, int track_every, int device);
  ~pss_impl() override;

  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override;
  // Adds the look-ahead general_work may need for its copy-out (fix for the reference's over-read,
  // SURVEY 8(a) quirk q1): peak_pos <= 9725 => frame_start + 9600 <= 18365 new items.
  void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;

  // The getters are polled from foreign threads while work() runs (GRC function probes at 2-10 Hz,
  // examples/rtlsdr_ltetrigger.grc:751-838).  The reference reads its members unsynchronised (lib/pss_impl.h:95-100);
  // here the scalars are relaxed atomics and the two rings sit behind a mutex work() takes for one store per half-frame.
  float max_psr() const override { return d_psr_max.load(std::memory_order_relaxed); }
  float mean_psr() const override { std::lock_guard<std::mutex> l(d_ring_mu); return ring_mean(d_psr_ring, d_psr_count); }
  float mean_cfo() const override { std::lock_guard<std::mutex> l(d_ring_mu); return ring_mean(d_cfo_ring, d_cfo_count); }
  void set_psr_threshold(float threshold) override { d_psr_threshold.store(threshold, std::memory_order_relaxed); }
  float psr_threshold() const override { return d_psr_threshold.load(std::memory_order_relaxed); }
  float tracking_score() const override { return (float)d_score.load(std::memory_order_relaxed); }

  static const pmt::pmt_t tracking_lost_tag_key;

private:
  static constexpr int slot_length = LT_SLOT_LEN;
  static constexpr int half_frame_length = LT_HALF_FRAME_LEN;
  static constexpr int symbol_sz = LT_SYMBOL_SZ;
  static constexpr int ring_size = LT_MOVING_AVG_SZ;
  static constexpr int max_lookahead = (LT_NLAGS - 1) - LT_SLOT_LEN; // largest frame_start

  static float ring_mean(const float *ring, size_t count);
  void score_hit();
  void score_miss();

  lt_pss *d_pss = nullptr;
  lt_cfo *d_cfo = nullptr;

  // tracking_t of the reference, flattened
  bool d_is_tracking = false;
  std::atomic<int> d_score{0};
  int d_timer = 0;
  bool d_tracking_lost = false;

  mutable std::mutex d_ring_mu;
  float d_psr_ring[ring_size] = {0};
  size_t d_psr_count = 0;
  float d_cfo_ring[ring_size] = {0};
  size_t d_cfo_count = 0;
  float d_psr = 0.f;
  std::atomic<float> d_psr_max{0.f};
  int d_peak_pos = 0;

  int d_N_id_2;
  std::atomic<float> d_psr_threshold;
  int d_track_after;
  int d_track_every;
};

} // namespace ltetrigger
} // namespace gr
