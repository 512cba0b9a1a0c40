(* gpr_hip.ml -- OCaml side of the gprhip drop-in: a module with the shape of
   [Gpr.Fitc_gp.Make_deriv (Spec).FITC] / [.Variational_FITC] for [Spec = Cov_se_iso.Deriv] and
   [Cov_se_fat.Deriv], whose arithmetic runs in libgprhip.so (include/gprhip.h) through gpr_hip_stubs.c.

   NOT COMPILED IN THIS REPOSITORY (no OCaml toolchain in the build image).  It is the source a maintainer of
   mmottl/gpr adds to lib/ (dune: (foreign_stubs (language c) (names gpr_hip_stubs)) (c_library_flags -lgprhip));
   the same call sequences are exercised, entry point by entry point, by the Python mirror gpr_amd/fitc_gp.py.

   Why the signature can be kept: in Interfaces.Sigs.Deriv the types Inputs.t, Model.t, Trained.t and hyper_t are
   abstract (lib/interfaces.ml:433, :459, :514, :895-899, :944-948).  Here they hold a handle to the device-resident
   problem plus what identifies an evaluation; only scalars, m-vectors and the gradient cross the boundary.
   Spec.Inputs.t / Spec.Inducing.t stay host Bigarrays (lib/cov_se_iso.mli:18-23).

   The caller's change is the functor line:
     bin/ocaml_gpr.ml:176    module GP = Gpr_hip.Se_fat            (was Fitc_gp.Make_deriv (Cov_se_fat.Deriv))
     test/save_data.ml:24    module GP = Gpr_hip.Se_iso            (was Fitc_gp.Make_deriv (Cov_se_iso.Deriv))
   and [GP.FITC], [GP.FIC], [GP.Variational_FITC], [GP.Variational_FIC] are then modules of type
   [Gpr.Interfaces.Sigs.Deriv with module Eval.Spec = Spec.Eval and module Deriv.Spec = Spec] (lib/fitc_gp.mli:83-135),
   so [GP.Variational_FIC.Deriv.Optim.Gsl.train] (bin/ocaml_gpr.ml:340-342) and the rest of the caller compile
   unchanged.  The devices the evaluation is sharded over are an argument of the functor ([Config.devices]): the
   library drives them from this one process (gprhip_ctx_* / gprhip_sharded_*, RCCL inside the library). *)

open Bigarray
open Lacaml.D

type problem (* custom block; its finaliser calls gprhip_problem_destroy *)

type hypers = {
  log_ell : float;
  log_sf2 : float;
  sigma2 : float;
  inducing : mat;
  tproj : mat option;
  variational : bool;
  model_only : bool;
  jitter : float;
  log_hetero_skedasticity : vec option;
  log_multiscales_m05 : mat option;
  reuse_v : bool;
}

external problem_create : int -> int -> int -> int -> int -> int -> int -> int -> problem
  = "gprhip_ml_problem_create_bc" "gprhip_ml_problem_create"
(* device cov_kind precision n big_d d m chunk_rows *)
external problem_destroy : problem -> unit = "gprhip_ml_problem_destroy"
external set_inputs : problem -> mat -> unit = "gprhip_ml_set_inputs"
external set_targets : problem -> vec -> unit = "gprhip_ml_set_targets"
external eval : problem -> hypers -> bool -> vec -> vec -> float * float * float * float * int = "gprhip_ml_eval"
(* want_grad grad coeffs -> (l1, l2, l, dl_dsigma2, n_hypers) *)
external n_hypers : problem -> int -> int = "gprhip_ml_n_hypers"
external predict : problem -> mat -> bool -> vec -> vec option -> unit = "gprhip_ml_predict"
external train_stats : problem -> vec option -> vec -> unit = "gprhip_ml_train_stats"
external covariances : problem -> mat -> int -> bool -> mat -> unit = "gprhip_ml_covariances"
external cov_samples : problem -> mat -> float -> float -> vec -> mat -> mat -> unit
  = "gprhip_ml_cov_samples_bc" "gprhip_ml_cov_samples"
external co_variance_coeffs : problem -> mat -> mat -> unit = "gprhip_ml_co_variance_coeffs"
external load_predictor : problem -> hypers -> vec option -> (mat * mat) option -> unit = "gprhip_ml_load_predictor"
external ar_len : problem -> int -> int = "gprhip_ml_ar_len"
external eval_pass1 : problem -> hypers -> bool -> int -> nativeint -> unit = "gprhip_ml_eval_pass1"
external eval_pass2 : problem -> nativeint -> nativeint -> unit = "gprhip_ml_eval_pass2"
external eval_finish : problem -> nativeint -> vec -> vec -> float * float * float * float * int
  = "gprhip_ml_eval_finish"
external sync : problem -> unit = "gprhip_ml_sync"
external stream : problem -> nativeint = "gprhip_ml_stream"
external set_timing : problem -> int -> unit = "gprhip_ml_set_timing"
external device_count : unit -> int = "gprhip_ml_device_count"
external version : unit -> string = "gprhip_ml_version"
external condition : problem -> float * float = "gprhip_ml_condition"
external debug_fetch : problem -> string -> vec -> unit = "gprhip_ml_debug_fetch"

(* single-process, multi-device (include/gprhip.h: gprhip_ctx_*, gprhip_sharded_*): the path the functors below take *)
type ctx (* custom block; finaliser calls gprhip_ctx_destroy (deferred by the library while sharded problems live) *)
type sharded (* custom block; finaliser calls gprhip_sharded_destroy *)

external ctx_create : int array -> ctx = "gprhip_ml_ctx_create"
external ctx_destroy : ctx -> unit = "gprhip_ml_ctx_destroy"
external ctx_ndev : ctx -> int = "gprhip_ml_ctx_ndev"
external ctx_comm_mode : ctx -> int = "gprhip_ml_ctx_comm_mode" (* 0 none, 1 RCCL, 2 same-device validation *)
external sharded_create : ctx -> int -> int -> int -> int -> int -> int -> int -> sharded
  = "gprhip_ml_sharded_create_bc" "gprhip_ml_sharded_create"
(* cov_kind precision n big_d d m chunk_rows *)
external sharded_destroy : sharded -> unit = "gprhip_ml_sharded_destroy"
external sharded_shard : sharded -> int -> int * int * int = "gprhip_ml_sharded_shard"
external shard_rows : int -> int -> int -> int * int = "gprhip_ml_shard_rows"
external sharded_problem : sharded -> int -> problem = "gprhip_ml_sharded_problem"
(* borrowed: valid while the sharded problem is reachable -- always kept in one record with it *)
external sharded_set_inputs : sharded -> mat -> unit = "gprhip_ml_sharded_set_inputs"
external sharded_set_targets : sharded -> vec -> unit = "gprhip_ml_sharded_set_targets"
external sharded_eval : sharded -> hypers -> bool -> vec -> vec -> float * float * float * float * int
  = "gprhip_ml_sharded_eval"
external sharded_predict : sharded -> mat -> bool -> vec -> vec option -> unit = "gprhip_ml_sharded_predict"
external sharded_train_stats : sharded -> vec option -> vec -> unit = "gprhip_ml_sharded_train_stats"
external sharded_comm_stats : sharded -> int * (int * int) * (float * float) = "gprhip_ml_sharded_comm_stats"
external sharded_set_timing : sharded -> int -> unit = "gprhip_ml_sharded_set_timing"

let cov_se_iso = 0
let cov_se_fat = 1
let f64 = 0
let f32_bulk = 1

(* What a covariance spec must tell the device path: which kernel, how a kernel value maps to [hypers], and how a
   position of the device gradient (the reference's Hyper.get_all order) is found for a hyper. *)
module type Device_spec = sig
  module Deriv : Gpr.Interfaces.Specs.Deriv with type Eval.Inducing.t = mat and type Eval.Inputs.t = mat

  val cov_kind : int
  val kernel_dim : Deriv.Eval.Kernel.t -> big_d:int -> int
  val hypers_of_kernel : Deriv.Eval.Kernel.t -> inducing:mat -> sigma2:float -> hypers
  val flags_of_kernel : Deriv.Eval.Kernel.t -> int
  val index_of_hyper : Deriv.Eval.Kernel.t -> inducing:mat -> Deriv.Hyper.t -> int
end

module Iso_spec : Device_spec with module Deriv = Gpr.Cov_se_iso.Deriv = struct
  module Deriv = Gpr.Cov_se_iso.Deriv

  let cov_kind = cov_se_iso
  let kernel_dim _ ~big_d = big_d

  let hypers_of_kernel k ~inducing ~sigma2 =
    let p = Deriv.Eval.Kernel.get_params k in
    {
      log_ell = p.Gpr.Cov_se_iso.Params.log_ell;
      log_sf2 = p.Gpr.Cov_se_iso.Params.log_sf2;
      sigma2;
      inducing;
      tproj = None;
      variational = false;
      model_only = false;
      jitter = !Gpr.Utils.cholesky_jitter;
      log_hetero_skedasticity = None;
      log_multiscales_m05 = None;
      reuse_v = false;
    }

  let flags_of_kernel _ = 0

  (* lib/cov_se_iso.ml:188-202: [Log_ell; Log_sf2; (ind = 1, dim = 1..d); (ind = 2, ...); ...], 0-based here *)
  let index_of_hyper _ ~inducing = function
    | `Log_ell -> 0
    | `Log_sf2 -> 1
    | `Inducing_hyper { Gpr.Cov_se_iso.ind; dim } -> 2 + ((ind - 1) * Mat.dim1 inducing) + (dim - 1)
end

module Fat_spec : Device_spec with module Deriv = Gpr.Cov_se_fat.Deriv = struct
  module Deriv = Gpr.Cov_se_fat.Deriv
  module P = Gpr.Cov_se_fat.Params

  let cov_kind = cov_se_fat
  let params k = (Deriv.Eval.Kernel.get_params k :> P.params)
  let kernel_dim k ~big_d:_ = (params k).P.d

  let hypers_of_kernel k ~inducing ~sigma2 =
    let p = params k in
    {
      log_ell = 0.;
      log_sf2 = p.P.log_sf2;
      sigma2;
      inducing;
      tproj = p.P.tproj;
      variational = false;
      model_only = false;
      jitter = !Gpr.Utils.cholesky_jitter;
      log_hetero_skedasticity = p.P.log_hetero_skedasticity;
      log_multiscales_m05 = p.P.log_multiscales_m05;
      reuse_v = false;
    }

  let flags_of_kernel k =
    let p = params k in
    (if p.P.tproj = None then 0 else 1)
    lor (if p.P.log_hetero_skedasticity = None then 0 else 2)
    lor if p.P.log_multiscales_m05 = None then 0 else 4

  (* lib/cov_se_fat.ml:290-342: [Log_sf2; inducing (ind-major); Proj (big_dim-major); hetero 1..m; multiscale (ind-major)] *)
  let index_of_hyper k ~inducing hyper =
    let p = params k in
    let d = Mat.dim1 inducing and m = Mat.dim2 inducing in
    let n_proj = match p.P.tproj with None -> 0 | Some t -> Mat.dim1 t * d in
    let n_het = match p.P.log_hetero_skedasticity with None -> 0 | Some _ -> m in
    match hyper with
    | `Log_sf2 -> 0
    | `Inducing_hyper { Gpr.Cov_se_fat.Inducing_hyper.ind; dim } -> 1 + ((ind - 1) * d) + (dim - 1)
    | `Proj { Gpr.Cov_se_fat.Proj_hyper.big_dim; small_dim } -> 1 + (d * m) + ((big_dim - 1) * d) + (small_dim - 1)
    | `Log_hetero_skedasticity i -> 1 + (d * m) + n_proj + (i - 1)
    | `Log_multiscale_m05 { Gpr.Cov_se_fat.Inducing_hyper.ind; dim } ->
        1 + (d * m) + n_proj + n_het + ((ind - 1) * d) + (dim - 1)
end



(* Where and how the device evaluation runs: an argument of the functor, fixed at functor application like
   Utils.cholesky_jitter is (lib/fitc_gp.ml:33). *)
module type Config = sig
  val devices : int array (* HIP devices the training points are row-sharded over; [| 0 |] = one GPU *)
  val precision : int (* f64 (reference parity) or f32_bulk *)
  val chunk_rows : int (* 0 = library default *)
end

module Default_config : Config = struct
  let devices = [| 0 |]
  let precision = f64
  let chunk_rows = 0
end

module All_devices_config : Config = struct
  let devices = Array.init (max 1 (device_count ())) (fun i -> i)
  let precision = f64
  let chunk_rows = 0
end

(* One evaluation on the device: everything Model / Trained / hyper_t expose is read off this record. *)
type evaluation = { l1 : float; l2 : float; l : float; dl_dsigma2 : float; grad : vec; coeffs : vec }

(* Device-resident copy of one set of inputs, sharded over the context's devices.  [first] is shard 0's device problem
   (borrowed from [sp]): the m x m model state is replicated on every device, so prediction, covariances and the
   export of the factors run there. *)
type device_inputs = { sp : sharded; first : problem; mutable owner : Obj.t option }

module Make_variant
    (S : Device_spec)
    (C : Config)
    (V : sig
      val variational : bool
      val fic : bool (* FIC_covariances instead of FITC_covariances: the only place the families differ (lib/fitc_gp.ml:565-627) *)
      val loc : string
    end) :
  Gpr.Interfaces.Sigs.Deriv with module Eval.Spec = S.Deriv.Eval and module Deriv.Spec = S.Deriv = struct
  module Espec = S.Deriv.Eval
  module Dspec = S.Deriv

  let context = lazy (ctx_create C.devices)
  let jitter = !Gpr.Utils.cholesky_jitter (* read once, at functor application: lib/fitc_gp.ml:33 *)

  let upload points ~kernel ~inducing_points =
    let big_d = Mat.dim1 points and n = Mat.dim2 points in
    let d = S.kernel_dim kernel ~big_d and m = Mat.dim2 inducing_points in
    if Mat.dim1 inducing_points <> d then
      failwith "Gpr_hip.Inputs.calc: dimension of inducing points disagrees with the kernel space";
    let sp = sharded_create (Lazy.force context) S.cov_kind C.precision n big_d d m C.chunk_rows in
    sharded_set_inputs sp points;
    { sp; first = sharded_problem sp 0; owner = None }

  (* ---------------------------------------------------------------- Eval ---------------------------------------- *)
  module Eval = struct
    module Spec = Espec

    module Inducing = struct
      type t = { kernel : Spec.Kernel.t; points : Spec.Inducing.t }

      let check_n_inducing ~n_inducing inputs =
        let n_inputs = Spec.Inputs.get_n_points inputs in
        if n_inputs < 1 || n_inducing > n_inputs then
          failwith
            (Printf.sprintf "Gpr.Fitc_gp.Make_common.check_n_inducing: violating 1 <= n_inducing (%d) <= n_inputs (%d)"
               n_inducing n_inputs)

      (* lib/fitc_gp.ml:62-92: choose columns, then the spec's create_inducing *)
      let choose kernel inputs indexes = Spec.Inputs.create_inducing kernel (Spec.Inputs.choose_subset inputs indexes)

      let iota n =
        let v = Gpr.Utils.Int_vec.create n in
        for i = 1 to n do
          v.{i} <- i
        done;
        v

      let choose_n_first_inputs kernel inputs ~n_inducing =
        check_n_inducing ~n_inducing inputs;
        choose kernel inputs (iota n_inducing)

      (* the first n_inducing steps of the reference's shuffle of the column indexes, drawing
         [Random.State.int rnd_state (n_inputs - i + 1)] at step i exactly as lib/fitc_gp.ml:74-92 does, so that a given
         random state selects the same inducing inputs *)
      let choose_n_random_inputs ?(rnd_state = Random.State.default) kernel inputs ~n_inducing =
        check_n_inducing ~n_inducing inputs;
        let n_inputs = Spec.Inputs.get_n_points inputs in
        let indexes = iota n_inputs in
        for i = 1 to n_inducing do
          let j = 1 + Random.State.int rnd_state (n_inputs - i + 1) in
          let at_j = indexes.{j} in
          indexes.{j} <- indexes.{i};
          indexes.{i} <- at_j
        done;
        choose kernel inputs (Gpr.Utils.Int_vec.sub indexes 1 n_inducing)

      let calc kernel points = { kernel; points }
      let get_points t = t.points
    end

    (* prepared inputs: the points and what they were prepared against; the device copy is made on first use, so
       points that are only predicted at are shipped per call by the library instead of occupying HBM *)
    module Inputs = struct
      type t = { inducing : Inducing.t; points : Spec.Inputs.t; dev : device_inputs Lazy.t }

      let create_default_kernel points ~n_inducing =
        Spec.Kernel.create (Spec.Inputs.create_default_kernel_params points ~n_inducing)

      let calc points (inducing : Inducing.t) =
        { inducing; points;
          dev = lazy (upload points ~kernel:inducing.Inducing.kernel ~inducing_points:inducing.Inducing.points) }

      let get_points t = t.points
    end

    module Input = struct
      type t = Inputs.t (* one column *)

      let calc inducing (point : Spec.Input.t) = Inputs.calc (Spec.Inputs.create [| point |]) inducing
    end

    (* run one evaluation on the device for (inputs, sigma2, targets) and record who owns the state it leaves *)
    let run (inputs : Inputs.t) ~sigma2 ~targets ~want_grad ~reuse_v owner =
      let dev = Lazy.force inputs.Inputs.dev in
      let ind = inputs.Inputs.inducing in
      let h = S.hypers_of_kernel ind.Inducing.kernel ~inducing:ind.Inducing.points ~sigma2 in
      let h = { h with variational = V.variational; model_only = targets = None; reuse_v; jitter } in
      (match targets with Some y -> sharded_set_targets dev.sp y | None -> ());
      let nh = n_hypers dev.first (S.flags_of_kernel ind.Inducing.kernel) in
      let grad = Vec.create (max nh 1) and coeffs = Vec.create (Mat.dim2 ind.Inducing.points) in
      dev.owner <- None;
      (* nobody owns a half-overwritten state if this raises *)
      let l1, l2, l, dl_dsigma2, _ = sharded_eval dev.sp h want_grad grad coeffs in
      dev.owner <- Some (Obj.repr owner);
      { l1; l2; l; dl_dsigma2; grad; coeffs }

    module Model = struct
      type t = { inputs : Inputs.t; sigma2 : float; reused : bool; mutable ev : evaluation option }
      type co_variance_coeffs = mat * mat (* (chol_km, r_mat), lib/fitc_gp.ml:240 *)

      let check_sigma2 sigma2 = if sigma2 < 0. then failwith "Model.check_sigma2: sigma2 < 0"

      let calc inputs ~sigma2 =
        check_sigma2 sigma2;
        { inputs; sigma2; reused = false; ev = None }

      let owns_state t =
        Lazy.is_val t.inputs.Inputs.dev
        && match (Lazy.force t.inputs.Inputs.dev).owner with Some o -> o == Obj.repr t | None -> false

      (* Model.update_sigma2 (lib/fitc_gp.ml:234-236): K_nm, V and r stay on the device when the last evaluation
         there was this model's *)
      let update_sigma2 t sigma2 =
        check_sigma2 sigma2;
        { t with sigma2; reused = owns_state t; ev = None }

      let evaluation ?(want_grad = false) t =
        match t.ev with
        | Some ev when (not want_grad) || Vec.dim ev.grad > 1 -> ev
        | _ ->
            let ev = run t.inputs ~sigma2:t.sigma2 ~targets:None ~want_grad ~reuse_v:t.reused t in
            t.ev <- Some ev;
            ev

      let ensure_state t = if not (owns_state t) then t.ev <- Some (run t.inputs ~sigma2:t.sigma2 ~targets:None ~want_grad:false ~reuse_v:false t)
      let calc_log_evidence t = (evaluation t).l1

      let calc_co_variance_coeffs t =
        ensure_state t;
        let m = Mat.dim2 t.inputs.Inputs.inducing.Inducing.points in
        let u = Mat.create m m and r = Mat.create m m in
        co_variance_coeffs (Lazy.force t.inputs.Inputs.dev).first u r;
        (u, r)

      let get_kernel t = t.inputs.Inputs.inducing.Inducing.kernel
      let get_sigma2 t = t.sigma2
      let get_inputs t = t.inputs
      let get_inducing t = t.inputs.Inputs.inducing
    end

    module Trained = struct
      type t = { model : Model.t; targets : vec; mutable ev : evaluation option }

      let calc (model : Model.t) ~targets =
        if Vec.dim targets <> Spec.Inputs.get_n_points model.Model.inputs.Inputs.points then
          failwith "Trained.calc: Vec.dim targets <> n";
        { model; targets; ev = None }

      let owns_state t =
        let inputs = t.model.Model.inputs in
        Lazy.is_val inputs.Inputs.dev
        && match (Lazy.force inputs.Inputs.dev).owner with Some o -> o == Obj.repr t | None -> false

      let evaluation ?(want_grad = false) t =
        match t.ev with
        | Some ev when (not want_grad) || Vec.dim ev.grad > 1 -> ev
        | _ ->
            let m = t.model in
            let ev =
              run m.Model.inputs ~sigma2:m.Model.sigma2 ~targets:(Some t.targets) ~want_grad ~reuse_v:m.Model.reused t
            in
            t.ev <- Some ev;
            ev

      let ensure_state t =
        if not (owns_state t) then
          let m = t.model in
          t.ev <-
            Some (run m.Model.inputs ~sigma2:m.Model.sigma2 ~targets:(Some t.targets) ~want_grad:false ~reuse_v:false t)

      let problem t = (Lazy.force t.model.Model.inputs.Inputs.dev).first
      let calc_mean_coeffs t = (evaluation t).coeffs
      let calc_log_evidence t = (evaluation t).l
      let get_model t = t.model
      let get_targets t = t.targets
    end

    (* Stats.calc (lib/fitc_gp.ml:304-374): the residual sums come back from the devices (per shard: sum, sum, max,
       sum), the ratios are formed here *)
    module Stats = struct
      type t = {
        n_samples : int;
        target_variance : float;
        sse : float;
        mse : float;
        rmse : float;
        smse : float;
        msll : float;
        mad : float;
        maxad : float;
      }

      let sums (trained : Trained.t) =
        Trained.ensure_state trained;
        let dev = Lazy.force trained.Trained.model.Model.inputs.Inputs.dev in
        let acc = Vec.create 4 in
        sharded_train_stats dev.sp None acc;
        acc

      let calc_n_samples (trained : Trained.t) = Vec.dim trained.Trained.targets

      let calc_target_variance (trained : Trained.t) =
        let y = trained.Trained.targets in
        let n = float (Vec.dim y) in
        let mean = Vec.sum y /. n in
        (Vec.sqr_nrm2 y /. n) -. (mean *. mean)

      let calc_sse trained = (sums trained).{1}
      let calc_mse trained = calc_sse trained /. float (calc_n_samples trained)
      let calc_rmse trained = sqrt (calc_mse trained)
      let calc_smse trained = calc_mse trained /. calc_target_variance trained

      let msll_of trained ~target_variance =
        let n = float (calc_n_samples trained) in
        (-.Trained.calc_log_evidence trained /. n) -. (0.5 *. (log (2. *. Float.pi *. target_variance) +. 1.))

      let calc_msll trained = msll_of trained ~target_variance:(calc_target_variance trained)
      let calc_mad trained = (sums trained).{2} /. float (calc_n_samples trained)
      let calc_maxad trained = (sums trained).{3}

      let calc trained =
        let s = sums trained in
        let n_samples = calc_n_samples trained in
        let f_n = float n_samples in
        let target_variance = calc_target_variance trained in
        let sse = s.{1} in
        let mse = sse /. f_n in
        {
          n_samples; target_variance; sse; mse; rmse = sqrt mse; smse = mse /. target_variance;
          msll = msll_of trained ~target_variance; mad = s.{2} /. f_n; maxad = s.{3};
        }
    end

    (* Predictors: either the state a trained model / model left on the device, or stored numbers (the [test] flow of
       bin/ocaml_gpr.ml:373-413) installed with gprhip_load_predictor into the problem of the inputs predicted at. *)
    module Mean_predictor = struct
      type t = Of_trained of Trained.t | Stored of { inducing : Spec.Inducing.t; coeffs : vec }

      let calc inducing ~coeffs =
        if Spec.Inducing.get_n_points inducing <> Vec.dim coeffs then
          failwith "Mean_predictor.calc: number of inducing points disagrees with dimension of coefficients";
        Stored { inducing; coeffs }

      let calc_trained trained = Of_trained trained

      let get_inducing = function
        | Of_trained t -> t.Trained.model.Model.inputs.Inputs.inducing.Inducing.points
        | Stored s -> s.inducing

      let get_coeffs = function Of_trained t -> Trained.calc_mean_coeffs t | Stored s -> s.coeffs
    end

    module Co_variance_predictor = struct
      type t =
        | Of_model of Model.t
        | Stored of { kernel : Spec.Kernel.t; inducing : Spec.Inducing.t; coeffs : Model.co_variance_coeffs }

      let calc kernel inducing coeffs = Stored { kernel; inducing; coeffs }
      let calc_model model = Of_model model
    end

    (* the device problem that serves predictions at [inputs] from stored numbers: that of [inputs] itself (its training
       rows are never evaluated -- the library only needs its kernel-space geometry and its streams) *)
    let load_stored (inputs : Inputs.t) ~sigma2 ?coeffs ?factors () =
      let dev = Lazy.force inputs.Inputs.dev in
      let ind = inputs.Inputs.inducing in
      let h = { (S.hypers_of_kernel ind.Inducing.kernel ~inducing:ind.Inducing.points ~sigma2) with jitter } in
      dev.owner <- None;
      load_predictor dev.first h coeffs factors;
      dev.first

    let same_inducing a b name =
      if a != b then failwith (name ^ ": predictor and inputs disagree about inducing points") (* lib/fitc_gp.ml:419-424 *)

    module Means = struct
      type t = { points : Spec.Inputs.t; means : vec }

      let calc mean_predictor (inputs : Inputs.t) =
        same_inducing (Mean_predictor.get_inducing mean_predictor) inputs.Inputs.inducing.Inducing.points "Means.calc";
        let means = Vec.create (Spec.Inputs.get_n_points inputs.Inputs.points) in
        (match mean_predictor with
        | Mean_predictor.Of_trained trained ->
            (* the trained model's state is replicated on every device: the test points are split over all of them *)
            Trained.ensure_state trained;
            sharded_predict (Lazy.force trained.Trained.model.Model.inputs.Inputs.dev).sp inputs.Inputs.points false means None
        | Mean_predictor.Stored s -> predict (load_stored inputs ~sigma2:0. ~coeffs:s.coeffs ()) inputs.Inputs.points false means None);
        { points = inputs.Inputs.points; means }

      let get t = t.means
    end

    module Mean = struct
      type t = float

      let calc mean_predictor (input : Input.t) = (Means.get (Means.calc mean_predictor input)).{1}
      let get t = t
    end

    module Variances = struct
      type t = { points : Spec.Inputs.t; variances : vec; sigma2 : float }

      let calc co_variance_predictor ~sigma2 (inputs : Inputs.t) =
        let nt = Spec.Inputs.get_n_points inputs.Inputs.points in
        let means = Vec.create nt and variances = Vec.create nt in
        (match co_variance_predictor with
        | Co_variance_predictor.Of_model model ->
            same_inducing model.Model.inputs.Inputs.inducing.Inducing.points inputs.Inputs.inducing.Inducing.points
              "Variances.calc";
            Model.ensure_state model;
            sharded_predict (Lazy.force model.Model.inputs.Inputs.dev).sp inputs.Inputs.points false means (Some variances)
        | Co_variance_predictor.Stored s ->
            same_inducing s.inducing inputs.Inputs.inducing.Inducing.points "Variances.calc";
            predict (load_stored inputs ~sigma2 ~factors:s.coeffs ()) inputs.Inputs.points false means (Some variances));
        { points = inputs.Inputs.points; variances; sigma2 }

      (* lib/fitc_gp.ml:487-496: at the model's own inputs the same numbers Variances.calc gives there *)
      let calc_model_inputs (model : Model.t) =
        calc (Co_variance_predictor.calc_model model) ~sigma2:model.Model.sigma2 model.Model.inputs

      let get ?(predictive = true) t = if predictive then Vec.add_const t.sigma2 t.variances else t.variances
    end

    module Variance = struct
      type t = { variance : float; sigma2 : float }

      let calc co_variance_predictor ~sigma2 (input : Input.t) =
        let v = Variances.calc co_variance_predictor ~sigma2 input in
        { variance = v.Variances.variances.{1}; sigma2 }

      let get ?(predictive = true) t = if predictive then t.variance +. t.sigma2 else t.variance
    end

    (* FITC_covariances / FIC_covariances (lib/fitc_gp.ml:565-627), upper triangle as the reference defines it *)
    module Covariances = struct
      type t = { points : Spec.Inputs.t; covariances : mat; sigma2 : float }

      let calc co_variance_predictor ~sigma2 (inputs : Inputs.t) =
        let nt = Spec.Inputs.get_n_points inputs.Inputs.points in
        let cov = Mat.create nt nt in
        let p =
          match co_variance_predictor with
          | Co_variance_predictor.Of_model model ->
              Model.ensure_state model;
              (Lazy.force model.Model.inputs.Inputs.dev).first
          | Co_variance_predictor.Stored s -> load_stored inputs ~sigma2 ~factors:s.coeffs ()
        in
        covariances p inputs.Inputs.points (if V.fic then 1 else 0) false cov;
        { points = inputs.Inputs.points; covariances = cov; sigma2 }

      let calc_model_inputs (model : Model.t) =
        calc (Co_variance_predictor.calc_model model) ~sigma2:model.Model.sigma2 model.Model.inputs

      let get ?(predictive = true) t =
        if not predictive then t.covariances
        else begin
          let res = lacpy ~uplo:`U t.covariances in
          for i = 1 to Mat.dim1 res do
            res.{i, i} <- res.{i, i} +. t.sigma2
          done;
          res
        end

      let get_variances t =
        { Variances.points = t.points; variances = Mat.copy_diag t.covariances; sigma2 = t.sigma2 }
    end

    (* Common_sampler / Common_cov_sampler (lib/fitc_gp.ml:629-697); the normal draws are GSL's, as in the reference *)
    module Sampler = struct
      type t = { mean : float; stddev : float }

      let calc ?(predictive = true) mean variance =
        let used_variance = Variance.get ~predictive variance in
        if used_variance < 0. then failwith (V.loc ^ ".Sampler.calc: negative variance");
        { mean = Mean.get mean; stddev = sqrt used_variance }

      let sample ?(rng = Gpr.Utils.default_rng) t = t.mean +. Gsl.Randist.gaussian_ziggurat rng ~sigma:t.stddev
      let samples ?rng t ~n = Vec.init n (fun _ -> sample ?rng t)
    end

    module Cov_sampler = struct
      type t = { means : vec; covariances : mat; add_diag : float; problem : problem Lazy.t }

      let calc ?(predictive = true) (means : Means.t) (covariances : Covariances.t) =
        if Vec.dim means.Means.means <> Mat.dim1 covariances.Covariances.covariances then
          failwith (V.loc ^ ".Cov_sampler.calc: means and covariances disagree about their dimension");
        {
          means = means.Means.means;
          covariances = covariances.Covariances.covariances;
          add_diag = (if predictive then covariances.Covariances.sigma2 else 0.);
          (* the factorisation and the product run on the first device of the context; any problem there serves *)
          problem = lazy (problem_create C.devices.(0) S.cov_kind f64 1 1 1 1 0);
        }

      let samples ?(rng = Gpr.Utils.default_rng) t ~n =
        let nt = Vec.dim t.means in
        let z = Mat.init_cols nt n (fun _ _ -> Gsl.Randist.gaussian_ziggurat rng ~sigma:1.) in
        let res = Mat.create nt n in
        cov_samples (Lazy.force t.problem) t.covariances t.add_diag jitter t.means z res;
        res

      let sample ?rng t = Mat.col (samples ?rng t ~n:1) 1
    end
  end

  (* ---------------------------------------------------------------- Deriv --------------------------------------- *)
  module Deriv = struct
    module Spec = Dspec

    module Inducing = struct
      type t = Eval.Inducing.t

      let calc = Eval.Inducing.calc
      let calc_eval t = t
    end

    module Inputs = struct
      type t = Eval.Inputs.t

      let calc inducing points = Eval.Inputs.calc points inducing
      let calc_eval t = t
    end

    let hyper_entry (ev : evaluation) (inducing : Eval.Inducing.t) hyper =
      ev.grad.{1 + S.index_of_hyper inducing.Eval.Inducing.kernel ~inducing:inducing.Eval.Inducing.points hyper}

    module Model = struct
      type t = Eval.Model.t
      type hyper_t = Eval.Model.t (* the whole gradient comes out of one device evaluation *)

      let calc = Eval.Model.calc
      let update_sigma2 = Eval.Model.update_sigma2
      let calc_eval t = t
      let calc_log_evidence_sigma2 t = (Eval.Model.evaluation ~want_grad:true t).dl_dsigma2

      let prepare_hyper t =
        ignore (Eval.Model.evaluation ~want_grad:true t);
        t

      let calc_log_evidence t hyper =
        hyper_entry (Eval.Model.evaluation ~want_grad:true t) t.Eval.Model.inputs.Eval.Inputs.inducing hyper
    end

    module Trained = struct
      type t = Eval.Trained.t
      type hyper_t = Eval.Trained.t

      let calc = Eval.Trained.calc
      let calc_eval t = t
      let calc_log_evidence_sigma2 t = (Eval.Trained.evaluation ~want_grad:true t).dl_dsigma2

      let prepare_hyper t =
        ignore (Eval.Trained.evaluation ~want_grad:true t);
        t

      let calc_log_evidence t hyper =
        hyper_entry
          (Eval.Trained.evaluation ~want_grad:true t)
          t.Eval.Trained.model.Eval.Model.inputs.Eval.Inputs.inducing hyper
    end

    module Test = struct
      (* element-wise check of the SPEC's derivative matrices by finite differences (lib/fitc_gp.ml:1223-1396): it
         involves no engine, only Spec -- the reference's own implementation is the one to run.  (The device analogue,
         which contracts differences of the device's covariance matrices with the device's W, X and v, is
         tests/test_gpu_parity.py::test_gradient_factors_against_differences_of_the_device_covariances.) *)
      module Ref = Gpr.Fitc_gp.Make_deriv (S.Deriv)

      let check_deriv_hyper = Ref.FITC.Deriv.Test.check_deriv_hyper

      (* Test.self_test (lib/fitc_gp.ml:1398-1462) over the device evaluation: forward differences of the model and
         trained log evidence against the device gradient, the reference's eps and tol *)
      let self_test ?(eps = 1e-8) ?(tol = 1e-2) kernel inducing_points points ~sigma2 ~targets hyper =
        let evidences kernel inducing_points points sigma2 =
          let inputs = Inputs.calc (Inducing.calc kernel inducing_points) points in
          let model = Model.calc inputs ~sigma2 in
          let trained = Trained.calc model ~targets in
          (model, trained, Eval.Model.calc_log_evidence model, Eval.Trained.calc_log_evidence trained)
        in
        let model, trained, mev, tev = evidences kernel inducing_points points sigma2 in
        let check name ~before ~after ~deriv =
          let fd = (after -. before) /. eps in
          if abs_float (fd -. deriv) > tol then
            failwith
              (Printf.sprintf "Gpr.Fitc_gp.Make_deriv.Test.self_test: %s: finite difference (%f) and derivative (%f) differ by more than %f"
                 name fd deriv tol)
        in
        match hyper with
        | `Sigma2 ->
            let _, _, mev2, tev2 = evidences kernel inducing_points points (sigma2 +. eps) in
            check "model sigma2" ~before:mev ~after:mev2 ~deriv:(Model.calc_log_evidence_sigma2 model);
            check "trained sigma2" ~before:tev ~after:tev2 ~deriv:(Trained.calc_log_evidence_sigma2 trained)
        | `Hyper hyper ->
            let value = Spec.Hyper.get_value kernel inducing_points points hyper in
            let kernel2, inducing2, points2 =
              Spec.Hyper.set_values kernel inducing_points points [| hyper |] (Vec.make 1 (value +. eps))
            in
            let _, _, mev2, tev2 = evidences kernel2 inducing2 points2 sigma2 in
            check "model hyper" ~before:mev ~after:mev2 ~deriv:(Model.calc_log_evidence (Model.prepare_hyper model) hyper);
            check "trained hyper" ~before:tev ~after:tev2
              ~deriv:(Trained.calc_log_evidence (Trained.prepare_hyper trained) hyper)
    end

    (* The optimisers of the reference (lib/fitc_gp.ml:1467-2017): same entry points, defaults and parameter vector
       ([log sigma2; hypers] when sigma2 is learnt, gradient entry 0 scaled by sigma2), driving the device evaluation. *)
    module Optim = struct
      let get_sigma2 targets = function
        | None -> Vec.sqr_nrm2 targets /. float (Vec.dim targets)
        | Some sigma2 when sigma2 < 0. -> failwith (Printf.sprintf "Optim.get_sigma2: sigma2 < 0: %f" sigma2)
        | Some sigma2 -> sigma2

      let get_kernel_inducing ?kernel ?n_rand_inducing ~inputs = function
        | Some inducing ->
            let kernel =
              match kernel with
              | Some kernel -> kernel
              | None -> Eval.Inputs.create_default_kernel inputs ~n_inducing:(Spec.Eval.Inducing.get_n_points inducing)
            in
            (kernel, inducing)
        | None ->
            let n_inputs = Spec.Eval.Inputs.get_n_points inputs in
            let n_inducing =
              match n_rand_inducing with
              | None -> min (n_inputs / 10) 1000
              | Some n when n < 1 -> failwith (Printf.sprintf "Gpr.Fitc_gp.Optim.get_kernel_inducing: n_rand_inducing (%d) < 1" n)
              | Some n when n > n_inputs ->
                  failwith (Printf.sprintf "Gpr.Fitc_gp.Optim.get_kernel_inducing: n_rand_inducing (%d) > n_inputs (%d)" n n_inputs)
              | Some n -> n
            in
            let kernel =
              match kernel with Some kernel -> kernel | None -> Eval.Inputs.create_default_kernel inputs ~n_inducing
            in
            (kernel, Eval.Inducing.choose_n_random_inputs kernel inputs ~n_inducing)

      let get_hypers_vals kernel inducing points hypers =
        let hypers = match hypers with None -> Spec.Hyper.get_all kernel inducing points | Some hypers -> hypers in
        (hypers, Vec.init (Array.length hypers) (fun i1 -> Spec.Hyper.get_value kernel inducing points hypers.(i1 - 1)))

      (* one device evaluation for a parameter vector: the trained model, and (if asked) the gradient of the NEGATIVE
         log evidence in the optimiser's parameters -- the body of multim_f / multim_dcommon (lib/fitc_gp.ml:1601-1636) *)
      let evaluate ~learn_sigma2 ~hypers ~targets (kernel, inducing, inputs) ~sigma2 ~gradient =
        let deriv_inputs = Inputs.calc (Inducing.calc kernel inducing) inputs in
        let trained = Trained.calc (Model.calc deriv_inputs ~sigma2) ~targets in
        (match gradient with
        | None -> ()
        | Some (set : int -> float -> unit) ->
            let off = if learn_sigma2 then 1 else 0 in
            if learn_sigma2 then set 0 (-.Trained.calc_log_evidence_sigma2 trained *. sigma2);
            if Array.length hypers > 0 then begin
              let hyper_t = Trained.prepare_hyper trained in
              Array.iteri (fun i h -> set (off + i) (-.Trained.calc_log_evidence hyper_t h)) hypers
            end);
        trained

      (* Optim.calc_gradient (lib/fitc_gp.ml:1674-1694): gradient of the log evidence, 1-based *)
      let calc_gradient ~learn_sigma2 ~sigma2 ~hypers ~trained =
        let n_hypers = Array.length hypers in
        let off = if learn_sigma2 then 1 else 0 in
        let gradient = Vec.create (n_hypers + off) in
        if learn_sigma2 then gradient.{1} <- Trained.calc_log_evidence_sigma2 trained *. sigma2;
        if n_hypers > 0 then begin
          let hyper_t = Trained.prepare_hyper trained in
          Array.iteri (fun i h -> gradient.{off + i + 1} <- Trained.calc_log_evidence hyper_t h) hypers
        end;
        gradient

      module Gsl = struct
        exception Optim_exception of exn

        let ignore_report ~iter:_ _ = ()

        let train ?(step = 1e-1) ?(tol = 1e-1) ?(epsabs = 1e-1) ?(report_trained_model = ignore_report)
            ?(report_gradient_norm = ignore_report) ?kernel ?sigma2 ?inducing ?n_rand_inducing ?(learn_sigma2 = true)
            ?hypers ~inputs ~targets () =
          let sigma2 = get_sigma2 targets sigma2 in
          let kernel, inducing = get_kernel_inducing ?kernel ?n_rand_inducing ~inputs inducing in
          let hypers, hyper_vals = get_hypers_vals kernel inducing inputs hypers in
          let n_hypers = Array.length hypers in
          let off = if learn_sigma2 then 1 else 0 in
          let n_gsl = n_hypers + off in
          let x0 = Gsl.Vector.create n_gsl in
          if learn_sigma2 then x0.{0} <- log sigma2;
          for i = 1 to n_hypers do
            x0.{off + i - 1} <- hyper_vals.{i}
          done;
          let sigma2_ref = ref sigma2 in
          let update_hypers ~x =
            if learn_sigma2 then sigma2_ref := exp x.{0};
            Spec.Hyper.set_values kernel inducing inputs hypers (Vec.init n_hypers (fun i -> x.{off + i - 1}))
          in
          let seen_exception = ref None in
          let guard f = try f () with exc -> seen_exception := Some exc; raise exc in
          let best = ref None and iter_count = ref 1 in
          let update_best trained log_evidence =
            match !best with
            | Some (_, old) when old >= log_evidence -> ()
            | _ ->
                report_trained_model ~iter:!iter_count trained;
                best := Some (trained, log_evidence)
          in
          let run ~x ~gradient =
            let trained = evaluate ~learn_sigma2 ~hypers ~targets (update_hypers ~x) ~sigma2:!sigma2_ref ~gradient in
            let l = Eval.Trained.calc_log_evidence trained in
            update_best trained l;
            -.l
          in
          let multim_f ~x = guard (fun () -> run ~x ~gradient:None) in
          let multim_fdf ~x ~g = guard (fun () -> run ~x ~gradient:(Some (fun i v -> g.{i} <- v))) in
          let multim_df ~x ~g = ignore (multim_fdf ~x ~g) in
          let module Gd = Gsl.Multimin.Deriv in
          let mumin = Gd.make Gd.VECTOR_BFGS2 n_gsl { Gsl.Fun.multim_f; multim_df; multim_fdf } ~x:x0 ~step ~tol in
          let g = Gsl.Vector.create n_gsl in
          let rec loop () =
            let neg_log_evidence = Gd.minimum ~x:x0 ~g mumin in
            (if Float.is_nan neg_log_evidence then
               match !seen_exception with
               | None -> failwith "Gpr.Optim.Gsl: optimization function returned nan"
               | Some exc -> raise (Optim_exception exc));
            let gnorm = Gsl.Blas.nrm2 g in
            (try report_gradient_norm ~iter:!iter_count gnorm with exc -> raise (Optim_exception exc));
            if gnorm < epsabs then match !best with Some (trained, _) -> trained | None -> assert false
            else begin
              incr iter_count;
              Gd.iterate mumin;
              loop ()
            end
          in
          loop ()
      end

      (* state shared by the two stochastic drivers *)
      type common = {
        learn_sigma2 : bool; hypers : Spec.Hyper.t array; targets : vec; inputs : Spec.Eval.Inputs.t;
        kernel : Spec.Eval.Kernel.t; inducing : Spec.Eval.Inducing.t; sigma2 : float; params : vec;
        trained : Trained.t; gradient : vec;
      }

      let common_create ?kernel ?sigma2 ?inducing ?n_rand_inducing ?(learn_sigma2 = true) ?hypers ~inputs ~targets () =
        let sigma2 = get_sigma2 targets sigma2 in
        let kernel, inducing = get_kernel_inducing ?kernel ?n_rand_inducing ~inputs inducing in
        let hypers, hyper_vals = get_hypers_vals kernel inducing inputs hypers in
        let off = if learn_sigma2 then 1 else 0 in
        let params = Vec.init (Array.length hypers + off) (fun i -> if learn_sigma2 && i = 1 then log sigma2 else hyper_vals.{i - off}) in
        let trained = evaluate ~learn_sigma2 ~hypers ~targets (kernel, inducing, inputs) ~sigma2 ~gradient:None in
        let gradient = calc_gradient ~learn_sigma2 ~sigma2 ~hypers ~trained in
        { learn_sigma2; hypers; targets; inputs; kernel; inducing; sigma2; params; trained; gradient }

      (* move to [params] (ascent direction already applied by the caller) and re-evaluate there *)
      let common_move c params =
        let off = if c.learn_sigma2 then 1 else 0 in
        let sigma2 = if c.learn_sigma2 then exp params.{1} else c.sigma2 in
        let kernel, inducing, inputs =
          Spec.Hyper.set_values c.kernel c.inducing c.inputs c.hypers
            (Vec.init (Array.length c.hypers) (fun i -> params.{off + i}))
        in
        let trained =
          evaluate ~learn_sigma2:c.learn_sigma2 ~hypers:c.hypers ~targets:c.targets (kernel, inducing, inputs) ~sigma2
            ~gradient:None
        in
        let gradient = calc_gradient ~learn_sigma2:c.learn_sigma2 ~sigma2 ~hypers:c.hypers ~trained in
        { c with kernel; inducing; inputs; sigma2; params; trained; gradient }

      let make_test step gradient_norm ?(epsabs = 0.1) ?max_iter ?(report = ignore) t =
        let max_iter = match max_iter with None -> -1 | Some n -> n in
        let rec loop t n =
          report t;
          if gradient_norm t < epsabs || n = max_iter then t else loop (step t) (n + 1)
        in
        loop t 0

      (* Optim.SGD (lib/fitc_gp.ml:1724-1826): eta_t = eta0 * tau / (tau + t), ascent on the log evidence *)
      module SGD = struct
        type t = { tau : float; eta0 : float; step : int; common : common }

        let create ?(tau = 0.1) ?(eta0 = 0.1) ?(step = 1) ?kernel ?sigma2 ?inducing ?n_rand_inducing ?learn_sigma2 ?hypers
            ~inputs ~targets () =
          { tau; eta0; step;
            common = common_create ?kernel ?sigma2 ?inducing ?n_rand_inducing ?learn_sigma2 ?hypers ~inputs ~targets () }

        let get_eta t = t.eta0 *. t.tau /. (t.tau +. float t.step)

        let step t =
          let params = copy t.common.params in
          axpy ~alpha:(get_eta t) t.common.gradient params;
          { t with step = t.step + 1; common = common_move t.common params }

        let gradient_norm t = nrm2 t.common.gradient
        let get_trained t = t.common.trained
        let get_step t = t.step
        let test ?epsabs ?max_iter ?report t = make_test step gradient_norm ?epsabs ?max_iter ?report t
      end

      (* Optim.SMD (lib/fitc_gp.ml:1835-2017): stochastic meta-descent.  Per-parameter gains eta adapted through the
         auxiliary vector nu; lambda * H nu is a CENTRAL difference of gradients along nu (:1952-1979: two more evaluations
         per step, at +eps and -eps), the nu update uses the OLD gains (:1992), and the hyper update keeps the reference's
         index convention as written (:1986-1990: [Vec.mul ~n:n_hypers eta ~ofsy:hyper_ix old_gradient] reads eta from its
         first entry while the gradient is read past the sigma2 slot).  Defaults and argument checks: :1848-1900. *)
      module SMD = struct
        type t = { eps : float; lambda : float; mu : float; eta : vec; nu : vec; common : common }

        let create ?(eps = 1e-8) ?lambda ?mu ?eta0 ?nu0 ?kernel ?sigma2 ?inducing ?n_rand_inducing ?learn_sigma2 ?hypers
            ~inputs ~targets () =
          let loc = "Gpr.Fitc_gp.Optim.SMD.create" in
          let lambda =
            match lambda with
            | None -> 0.1
            | Some l when l < 0. || l > 1. -> failwith (Printf.sprintf "%s: violating 0 <= lambda(%f) <= 1" loc l)
            | Some l -> l
          in
          let mu =
            match mu with
            | None -> 1e-3
            | Some mu when mu < 0. -> failwith (Printf.sprintf "%s: violating 0 <= mu(%f)" loc mu)
            | Some mu -> mu
          in
          let common = common_create ?kernel ?sigma2 ?inducing ?n_rand_inducing ?learn_sigma2 ?hypers ~inputs ~targets () in
          let n = Vec.dim common.params in
          let eta =
            match eta0 with
            | None -> Vec.make n 1e-3
            | Some v ->
                if Vec.dim v <> n then failwith (Printf.sprintf "%s: dim(eta0) = %d <> n_all_hypers(%d)" loc (Vec.dim v) n);
                for i = 1 to n do
                  if v.{i} <= 0. then failwith (Printf.sprintf "%s: eta0.{%d} < 0: %f" loc i v.{i})
                done;
                copy v
          in
          let nu =
            match nu0 with
            | None -> Vec.make n 1e-3
            | Some v ->
                if Vec.dim v <> n then failwith (Printf.sprintf "%s: dim(nu0) = %d <> n_all_hypers(%d)" loc (Vec.dim v) n);
                copy v
          in
          { eps; lambda; mu; eta; nu; common }

        let step t =
          let c = t.common in
          let n = Vec.dim c.params in
          let n_hypers = Array.length c.hypers in
          let off = if c.learn_sigma2 then 1 else 0 in
          let old_eta = t.eta and old_nu = t.nu and old_gradient = c.gradient in
          (* lambda * H nu ~ lambda / (2 eps) * (grad(params + eps nu) - grad(params - eps nu));  params.{1} is
             log sigma2 when it is learnt, so "exp (log_old_sigma2 +. eps *. old_nu.{1})" is the same shift *)
          let grad_at eps =
            let shifted = copy c.params in
            axpy ~alpha:eps old_nu shifted;
            copy (common_move c shifted).gradient
          in
          let lambda_hessian_nu = Vec.sub (grad_at t.eps) (grad_at (-.t.eps)) in
          scal (t.lambda /. (2. *. t.eps)) lambda_hessian_nu;
          let eta = Vec.init n (fun i -> old_eta.{i} *. Float.max 0.5 (1. +. (t.mu *. old_gradient.{i} *. old_nu.{i}))) in
          let params = copy c.params in
          if c.learn_sigma2 then params.{1} <- c.params.{1} +. (eta.{1} *. old_gradient.{1});
          for i = 1 to n_hypers do
            params.{off + i} <- c.params.{off + i} +. (eta.{i} *. old_gradient.{off + i})
          done;
          let nu = Vec.init n (fun i -> (old_eta.{i} *. (old_gradient.{i} +. lambda_hessian_nu.{i})) +. (t.lambda *. old_nu.{i})) in
          { t with eta; nu; common = common_move c params }

        let gradient_norm t = nrm2 t.common.gradient
        let get_trained t = t.common.trained
        let get_eta t = t.eta
        let get_nu t = t.nu
        let test ?epsabs ?max_iter ?report t = make_test step gradient_norm ?epsabs ?max_iter ?report t
      end
    end
  end
end

(* Fitc_gp.Make_deriv (lib/fitc_gp.mli:83-135): the four model families over one covariance spec.

   The sharing constraints of the reference's signature (lib/fitc_gp.mli:87-135: FIC.Eval.Model = FITC.Eval.Model,
   FIC.Eval.Means = FITC.Eval.Means, Variational_FITC.Eval.Inputs = FITC.Eval.Inputs ...) are NOT reproduced: each family
   is its own application of Make_variant, sealed with Sigs.Deriv, so its Inducing / Inputs / Model / Trained / Means types
   are abstract and distinct from the other families'.  Which of the reference's own programs that affects, precisely:
     - bin/ocaml_gpr.ml uses one family throughout (module FIC = GP.Variational_FIC.Eval, :177; the trainer of the same
       family, :340): type-checks as it is.  (:343 matches GP.FIC.Deriv.Optim.Gsl.Optim_exception, another family's
       exception constructor: legal here as there, and as there it never matches what Variational_FIC's trainer raises.)
     - test/test_derivatives.ml uses GP.FITC only (:22-60): type-checks as it is.
     - test/save_data.ml:136-146 does NOT type-check: it passes a FITC model to FIC.Covariances.calc_model_inputs (:136)
       and FITC means to FIC.Cov_sampler.calc (:138), i.e. it needs FIC.Eval.Model = FITC.Eval.Model and
       FIC.Eval.Means = FITC.Eval.Means.  With this module the FIC block of that program evaluates under its own family:
         let fic_inputs = FIC.Inputs.calc training_inputs (FIC.Inducing.calc kernel inducing_points) in
         let fic_model = FIC.Model.calc fic_inputs ~sigma2 in
         let fic_means = FIC.Means.calc (FIC.Mean_predictor.calc_trained (FIC.Trained.calc fic_model ~targets)) fic_inputs in
       (one more evaluation on the device, same numbers: the families differ in Covariances only, lib/fitc_gp.ml:565-627).
   Building FIC by [include]-ing FITC's modules (the reference's own construction: Make_FITC_deriv / Make_FIC_deriv over one Make_common_deriv, lib/fitc_gp.ml:2056-2108) needs
   Make_variant split into an unsealed base and two covariance flavours over it; without a compiler in the build image that
   refactoring of 1000 unchecked lines was judged more likely to break the module than to help a caller. *)
module Make_deriv (S : Device_spec) (C : Config) = struct
  module type Sig = Gpr.Interfaces.Sigs.Deriv with module Eval.Spec = S.Deriv.Eval and module Deriv.Spec = S.Deriv

  module FITC : Sig = Make_variant (S) (C) (struct let variational = false let fic = false let loc = "FITC" end)
  module FIC : Sig = Make_variant (S) (C) (struct let variational = false let fic = true let loc = "FIC" end)

  module Variational_FITC : Sig =
    Make_variant (S) (C) (struct let variational = true let fic = false let loc = "Variational_FITC" end)

  module Variational_FIC : Sig =
    Make_variant (S) (C) (struct let variational = true let fic = true let loc = "Variational_FIC" end)
end

(* the caller's functor line: [module GP = Gpr_hip.Se_fat] (bin/ocaml_gpr.ml:176), [module GP = Gpr_hip.Se_iso]
   (test/save_data.ml:24); [Se_*_all] shard over every device of the node *)
module Se_iso = Make_deriv (Iso_spec) (Default_config)
module Se_fat = Make_deriv (Fat_spec) (Default_config)
module Se_iso_all = Make_deriv (Iso_spec) (All_devices_config)
module Se_fat_all = Make_deriv (Fat_spec) (All_devices_config)

module Se_fat_f32 =
  Make_deriv
    (Fat_spec)
    (struct
      let devices = [| 0 |]
      let precision = f32_bulk
      let chunk_rows = 0
    end)
