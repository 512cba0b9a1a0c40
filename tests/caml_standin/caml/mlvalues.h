/* STAND-IN, NOT OCaml's header.  Declarations only -- just enough of the documented OCaml C interface ("Interfacing C
 * with OCaml", OCaml manual ch. 22: value representation and accessor macros) for tests/test_abi.py to run
 * `gcc -fsyntax-only` over bindings/gpr_hip_stubs.c in an image that has no OCaml installation.  Nothing here is ever
 * linked or run; a real build uses the compiler's own <caml/...> and must not see this directory. */
#ifndef GPRHIP_CAML_STANDIN_MLVALUES_H
#define GPRHIP_CAML_STANDIN_MLVALUES_H
#include <stddef.h>
#include <stdint.h>

typedef intptr_t intnat;
typedef uintptr_t uintnat;
typedef intnat value;
typedef uintnat mlsize_t;
typedef uintnat header_t;

#define Val_long(x) ((value)(((uintnat)(intnat)(x) << 1) + 1))
#define Long_val(x) ((intnat)(x) >> 1)
#define Val_int(x) Val_long(x)
#define Int_val(x) ((int)Long_val(x))
#define Val_unit Val_int(0)
#define Val_bool(x) Val_int((x) != 0)
#define Bool_val(x) Int_val(x)
#define Val_true Val_int(1)
#define Val_false Val_int(0)
#define Is_long(x) (((x) & 1) != 0)
#define Is_block(x) (((x) & 1) == 0)
#define Field(x, i) (((value*)(x))[i])
#define Hd_val(v) (((header_t*)(v))[-1])
#define Wosize_val(v) ((mlsize_t)(Hd_val(v) >> 10))
#define String_val(v) ((const char*)(v))
#define Bytes_val(v) ((unsigned char*)(v))
#define Data_custom_val(v) ((void*)&Field((v), 1))
#define Nativeint_val(v) (*((intnat*)Data_custom_val(v)))
#define Int64_val(v) (*((int64_t*)Data_custom_val(v)))
double caml_Double_val(value);
#define Double_val(v) caml_Double_val(v)
#define Double_flat_field(v, i) (((double*)(v))[i])
#define CAMLprim
#define CAMLextern extern
#endif
