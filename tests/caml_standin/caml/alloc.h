/* STAND-IN, NOT OCaml's header (see mlvalues.h in this directory): allocation functions the stubs call. */
#ifndef GPRHIP_CAML_STANDIN_ALLOC_H
#define GPRHIP_CAML_STANDIN_ALLOC_H
#include "mlvalues.h"
value caml_alloc(mlsize_t wosize, int tag);
value caml_alloc_tuple(mlsize_t n);
value caml_alloc_small(mlsize_t wosize, int tag);
value caml_copy_double(double d);
value caml_copy_string(const char* s);
value caml_copy_nativeint(intnat i);
value caml_copy_int64(int64_t i);
#endif
