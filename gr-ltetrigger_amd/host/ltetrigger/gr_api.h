// Where the GNU Radio block API comes from.  With a real GNU Radio (CMakeLists.txt sets LTETRIGGER_USE_GNURADIO when
// find_package(Gnuradio) succeeds) the blocks compile against its headers, as the reference's do
// (include/ltetrigger/pss.h:24-25, lib/pss_impl.h:24-28); otherwise against the in-repo stand-in with the same
// class and method names (host/runtime.h), because neither container of this project has GNU Radio.
#pragma once
#ifdef LTETRIGGER_USE_GNURADIO
#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <pmt/pmt.h>
#else
#include "../runtime.h"
#endif
