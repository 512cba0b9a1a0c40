/* STAND-IN, NOT OCaml's header (see mlvalues.h in this directory): exception-raising functions (they do not return). */
#ifndef GPRHIP_CAML_STANDIN_FAIL_H
#define GPRHIP_CAML_STANDIN_FAIL_H
#include "mlvalues.h"
void caml_failwith(const char* msg) __attribute__((noreturn));
void caml_invalid_argument(const char* msg) __attribute__((noreturn));
void caml_raise_out_of_memory(void) __attribute__((noreturn));
void caml_raise_not_found(void) __attribute__((noreturn));
#endif
