// ThreadSanitizer check of the host side's concurrent parts (built and run by tests/test_sanitizers.py; CPU only):
//   1. two threads post "track" / "drop" into one gr::ltetrigger::cellstore while a third polls tracking() / cells() /
//      latest_cell() (the reference's use: lib/cellstore_impl.cc:60-105, polled from the application's thread,
//      examples/cell_search_file.py:101-129);
//   2. a flowgraph under the thread-per-block scheduler (top_block::run_threaded): source -> head -> a block that tags
//      and publishes messages from work() -> two sinks, a message_debug and the cellstore fed from the block's thread
//      while the main thread polls them.  The same flowgraph under the single-threaded scheduler must see the same
//      items, tags and messages.
// Exit code 0 and no ThreadSanitizer report = pass.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../gr-ltetrigger_amd/host/cellstore_impl.h"
#include "../../gr-ltetrigger_amd/host/runtime.h"

namespace {

pmt::pmt_t cell(long id)
{
  pmt::pmt_t d = pmt::make_dict();
  d = pmt::dict_add(d, pmt::intern("cell_id"), pmt::from_long(id));
  d = pmt::dict_add(d, pmt::intern("nof_prb"), pmt::from_long(6));
  return d;
}

// every `period` items: a tag on the first item of the call and a "track" message; the cell before it is dropped
class tagger : public gr::sync_block
{
public:
  explicit tagger(int period)
      : gr::sync_block("tagger", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))), d_period(period)
  {
    message_port_register_out(pmt::intern("track"));
    message_port_register_out(pmt::intern("drop"));
    set_output_multiple(period);
  }
  int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
  {
    const gr_complex *x = static_cast<const gr_complex *>(in[0]);
    gr_complex *y = static_cast<gr_complex *>(out[0]);
    for (int i = 0; i < d_period; ++i) y[i] = x[i] * 2.0f;
    add_item_tag(0, nitems_written(0), pmt::intern("mark"), pmt::from_long(d_calls));
    pmt::pmt_t c = cell(d_calls % 504); // (a cell id: the cellstore rejects 1000, its "key missing" sentinel)
    message_port_pub(pmt::intern("track"), c);
    if (d_last) message_port_pub(pmt::intern("drop"), d_last);
    d_last = c;
    ++d_calls;
    (void)n;
    return d_period;
  }

private:
  int d_period;
  long d_calls = 0;
  pmt::pmt_t d_last;
};

class collector : public gr::sync_block
{
public:
  collector() : gr::sync_block("collector", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0)) {}
  int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
  {
    const gr_complex *x = static_cast<const gr_complex *>(in[0]);
    for (int i = 0; i < n; ++i) sum += (double)x[i].real();
    std::vector<gr::tag_t> t;
    get_tags_in_range(t, 0, nitems_read(0), nitems_read(0) + (uint64_t)n);
    tags += (long)t.size();
    items += n;
    return n;
  }
  double sum = 0;
  long tags = 0, items = 0;
};

struct Result {
  double sum_a, sum_b;
  long tags_a, tags_b, items_a, items_b;
  int msgs;
  size_t cells;
};

Result flowgraph(bool threaded, std::atomic<long> *polls)
{
  std::vector<gr_complex> data(4096);
  for (size_t i = 0; i < data.size(); ++i) data[i] = gr_complex((float)(i % 17), -(float)(i % 5));
  auto src = gr::blocks::vector_source_c::make(data, true);
  auto head = gr::blocks::head::make(960 * 400);
  auto tag = std::make_shared<tagger>(960);
  auto a = std::make_shared<collector>();
  auto b = std::make_shared<collector>();
  auto dbg = gr::blocks::message_debug::make();
  auto store = gr::ltetrigger::cellstore::make();
  // (on the heap: a std::mutex has no destructor call ThreadSanitizer could see, and the next call's top_block would land on
  // the same stack address -- reported as a lock of a destroyed mutex)
  auto tbp = std::make_unique<gr::top_block>();
  gr::top_block &tb = *tbp;
  tb.connect(src, head);
  tb.connect(head, tag);
  tb.connect(tag, a);
  tb.connect(tag, b);
  tb.msg_connect(tag, "track", dbg, "store");
  tb.msg_connect(tag, "track", store, "track");
  tb.msg_connect(tag, "drop", store, "drop");
  std::atomic<bool> stop{false};
  std::thread poller([&] {
    while (!stop.load()) {
      (void)store->tracking();
      (void)store->cells().size();
      (void)store->latest_cell();
      (void)dbg->num_messages();
      if (polls) polls->fetch_add(1);
      std::this_thread::yield();
    }
  });
  if (threaded) tb.run_threaded();
  else tb.run();
  stop.store(true);
  poller.join();
  return Result{a->sum, b->sum, a->tags, b->tags, a->items, b->items, dbg->num_messages(), store->cells().size()};
}

} // namespace

int main()
{
  // ---- 1. cellstore under two posting threads and a polling one
  {
    auto store = gr::ltetrigger::cellstore::make();
    std::atomic<bool> stop{false};
    std::atomic<long> seen{0};
    auto poster = [&](long base) {
      for (long i = 0; i < 2000; ++i) {
        pmt::pmt_t c = cell((base + i) % 504);
        store->post(pmt::intern("track"), c);
        if (i % 3) store->post(pmt::intern("drop"), c);
      }
    };
    std::thread p1(poster, 0L), p2(poster, 100000L);
    std::thread poll([&] {
      while (!stop.load()) {
        seen.fetch_add((long)store->cells().size() + (store->tracking() ? 1 : 0));
        (void)store->latest_cell();
      }
    });
    p1.join();
    p2.join();
    stop.store(true);
    poll.join();
    const size_t left = store->cells().size();
    if (left != 2 * 667) { // i % 3 == 0 for i in [0, 2000): 667 cells per poster stay
      std::fprintf(stderr, "cellstore: %zu cells left, expected %d\n", left, 2 * 667);
      return 2;
    }
  }
  // ---- 2. the two schedulers on one flowgraph
  std::atomic<long> polls{0};
  const Result s = flowgraph(false, nullptr);
  for (int rep = 0; rep < 3; ++rep) {
    const Result t = flowgraph(true, &polls);
    if (t.items_a != s.items_a || t.items_b != s.items_b || t.items_a != 960 * 400 || t.sum_a != s.sum_a || t.sum_b != s.sum_b || t.tags_a != s.tags_a ||
        t.tags_b != s.tags_b || t.tags_a != 400 || t.msgs != s.msgs || t.msgs != 400 || t.cells != s.cells || t.cells != 1) {
      std::fprintf(stderr, "threaded run %d differs: items %ld/%ld vs %ld/%ld, sums %.1f/%.1f vs %.1f/%.1f, tags %ld/%ld vs %ld/%ld, msgs %d vs %d, cells %zu vs %zu\n", rep,
                   t.items_a, t.items_b, s.items_a, s.items_b, t.sum_a, t.sum_b, s.sum_a, s.sum_b, t.tags_a, t.tags_b, s.tags_a, s.tags_b, t.msgs, s.msgs, t.cells, s.cells);
      return 3;
    }
  }
  std::printf("runtime_tsan ok: cellstore under 3 threads; thread-per-block run = single-threaded run (384000 items, 400 tags, 400 messages), %ld polls meanwhile\n", polls.load());
  return 0;
}
