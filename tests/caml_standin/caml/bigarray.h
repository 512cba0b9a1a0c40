/* STAND-IN, NOT OCaml's header (see mlvalues.h in this directory): the Bigarray descriptor and its accessors. */
#ifndef GPRHIP_CAML_STANDIN_BIGARRAY_H
#define GPRHIP_CAML_STANDIN_BIGARRAY_H
#include "mlvalues.h"
struct caml_ba_proxy;
struct caml_ba_array {
  void* data;
  intnat num_dims;
  intnat flags;
  struct caml_ba_proxy* proxy;
  intnat dim[1]; /* num_dims entries */
};
enum caml_ba_kind { CAML_BA_FLOAT32 = 0, CAML_BA_FLOAT64 = 1, CAML_BA_KIND_MASK = 0xFF };
enum caml_ba_layout { CAML_BA_C_LAYOUT = 0, CAML_BA_FORTRAN_LAYOUT = 0x100, CAML_BA_LAYOUT_MASK = 0x100 };
enum caml_ba_managed { CAML_BA_EXTERNAL = 0, CAML_BA_MANAGED = 0x200, CAML_BA_MAPPED_FILE = 0x400 };
#define Caml_ba_array_val(v) ((struct caml_ba_array*)Data_custom_val(v))
#define Caml_ba_data_val(v) (Caml_ba_array_val(v)->data)
value caml_ba_alloc(int flags, int num_dims, void* data, intnat* dim);
value caml_ba_alloc_dims(int flags, int num_dims, void* data, ... /* dimensions, as intnat */);
#endif
