#pragma once
// With GNU Radio present this would come from <gnuradio/attributes.h>; the in-repo runtime needs
// default visibility only.
#define LTETRIGGER_API __attribute__((visibility("default")))
