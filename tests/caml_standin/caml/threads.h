/* STAND-IN, NOT OCaml's header (see mlvalues.h in this directory): the master runtime lock. */
#ifndef GPRHIP_CAML_STANDIN_THREADS_H
#define GPRHIP_CAML_STANDIN_THREADS_H
void caml_release_runtime_system(void);
void caml_acquire_runtime_system(void);
#endif
