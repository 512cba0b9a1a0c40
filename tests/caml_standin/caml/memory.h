/* STAND-IN, NOT OCaml's header (see mlvalues.h in this directory).  The real CAMLparam / CAMLlocal macros register
 * local roots with the garbage collector; these only make the compiler check that every argument is an lvalue of type
 * `value` and that CAMLreturn's operand has the function's return type. */
#ifndef GPRHIP_CAML_STANDIN_MEMORY_H
#define GPRHIP_CAML_STANDIN_MEMORY_H
#include "mlvalues.h"
void caml_modify(value* fp, value v);
void caml_initialize(value* fp, value v);
#define Store_field(block, offset, val) caml_modify(&Field((block), (offset)), (val))
#define GPRHIP_STANDIN_ROOT(x) ((void)sizeof(*(value*)0 = *(&(x))), (void)(value*)&(x))
#define CAMLparam0() int caml__frame = 0; (void)caml__frame
#define CAMLparam1(a) CAMLparam0(); GPRHIP_STANDIN_ROOT(a)
#define CAMLparam2(a, b) CAMLparam1(a); GPRHIP_STANDIN_ROOT(b)
#define CAMLparam3(a, b, c) CAMLparam2(a, b); GPRHIP_STANDIN_ROOT(c)
#define CAMLparam4(a, b, c, d) CAMLparam3(a, b, c); GPRHIP_STANDIN_ROOT(d)
#define CAMLparam5(a, b, c, d, e) CAMLparam4(a, b, c, d); GPRHIP_STANDIN_ROOT(e)
#define CAMLxparam1(a) (void)caml__frame; GPRHIP_STANDIN_ROOT(a)
#define CAMLxparam2(a, b) CAMLxparam1(a); GPRHIP_STANDIN_ROOT(b)
#define CAMLxparam3(a, b, c) CAMLxparam2(a, b); GPRHIP_STANDIN_ROOT(c)
#define CAMLxparam4(a, b, c, d) CAMLxparam3(a, b, c); GPRHIP_STANDIN_ROOT(d)
#define CAMLxparam5(a, b, c, d, e) CAMLxparam4(a, b, c, d); GPRHIP_STANDIN_ROOT(e)
#define CAMLlocal1(x) value x = Val_unit
#define CAMLlocal2(x, y) value x = Val_unit, y = Val_unit
#define CAMLlocal3(x, y, z) value x = Val_unit, y = Val_unit, z = Val_unit
#define CAMLlocal4(x, y, z, t) value x = Val_unit, y = Val_unit, z = Val_unit, t = Val_unit
#define CAMLreturn(x) do { (void)caml__frame; return (x); } while (0)
#define CAMLreturn0 do { (void)caml__frame; return; } while (0)
#define CAMLreturnT(type, x) do { type caml__result = (x); (void)caml__frame; return caml__result; } while (0)
#endif
