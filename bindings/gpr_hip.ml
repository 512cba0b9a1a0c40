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
     test/save_data.ml:24    module GP = Gpr_hip.Se_iso            (was Fitc_gp.Make_deriv (Cov_se_iso.Deriv)) *)

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

(* One evaluation on the device: everything Model / Trained / hyper_t expose is read off this record. *)
type evaluation = { l1 : float; l2 : float; l : float; dl_dsigma2 : float; grad : vec; coeffs : vec }

module Make (S : Device_spec) (V : sig
  val variational : bool
  val precision : int
  val device : int
end) =
struct
  module Spec = S.Deriv.Eval

  module Inducing = struct
    type t = { kernel : Spec.Kernel.t; points : mat }

    let calc kernel points = { kernel; points }
    let get_points t = t.points
    let get_kernel t = t.kernel
    let calc_eval t = t
    let choose_n_first_inputs _kernel inputs ~n_inducing = lacpy ~n:n_inducing inputs

    let choose_n_random_inputs ?(rnd_state = Random.get_state ()) _kernel inputs ~n_inducing =
      (* as lib/fitc_gp.ml:76-99: a partial Fisher-Yates draw of column indices *)
      let n = Mat.dim2 inputs in
      let idx = Array.init n (fun i -> i + 1) in
      let res = Mat.create (Mat.dim1 inputs) n_inducing in
      for c = 1 to n_inducing do
        let r = c - 1 + Random.State.int rnd_state (n - c + 1) in
        let tmp = idx.(c - 1) in
        idx.(c - 1) <- idx.(r);
        idx.(r) <- tmp;
        ignore (copy (Mat.col inputs idx.(c - 1)) ~y:(Mat.col res c))
      done;
      res
  end

  module Inputs = struct
    (* the problem is created, and the training inputs uploaded, on first use: inputs that are only predicted at
       never occupy HBM *)
    type t = { inducing : Inducing.t; points : mat; problem : problem Lazy.t }

    let calc points (inducing : Inducing.t) =
      let big_d = Mat.dim1 points and n = Mat.dim2 points in
      let d = S.kernel_dim inducing.Inducing.kernel ~big_d and m = Mat.dim2 inducing.Inducing.points in
      if Mat.dim1 inducing.Inducing.points <> d then
        failwith "Gpr_hip.Inputs.calc: dimension of inducing points disagrees with the kernel space";
      let problem =
        lazy
          (let p = problem_create V.device S.cov_kind V.precision n big_d d m 0 in
           set_inputs p points;
           p)
      in
      { inducing; points; problem }

    let get_points t = t.points
    let calc_eval t = t
    let create_default_kernel points ~n_inducing = Spec.Inputs.create_default_kernel_params points ~n_inducing |> Spec.Kernel.create
  end

  (* the device holds the state of the last evaluation only: who that was *)
  let state_owner : (problem * Obj.t) option ref = ref None

  let run (inputs : Inputs.t) ~sigma2 ~targets ~want_grad ~reuse_v owner =
    let p = Lazy.force inputs.Inputs.problem in
    let ind = inputs.Inputs.inducing in
    let h = S.hypers_of_kernel ind.Inducing.kernel ~inducing:ind.Inducing.points ~sigma2 in
    let h = { h with variational = V.variational; model_only = targets = None; reuse_v } in
    (match targets with Some y -> set_targets p y | None -> ());
    let nh = n_hypers p (S.flags_of_kernel ind.Inducing.kernel) in
    let grad = Vec.create (max nh 1) and coeffs = Vec.create (Mat.dim2 ind.Inducing.points) in
    state_owner := None;
    let l1, l2, l, dl_dsigma2, _ = eval p h want_grad grad coeffs in
    state_owner := Some (p, Obj.repr owner);
    { l1; l2; l; dl_dsigma2; grad; coeffs }

  module Model = struct
    type t = { inputs : Inputs.t; sigma2 : float; ev : evaluation Lazy.t; reused : bool }
    type co_variance_coeffs = mat * mat
    type hyper_t = t

    let rec make ?(reused = false) inputs sigma2 =
      if sigma2 < 0. then failwith "Model.check_sigma2: sigma2 < 0";
      let rec t = { inputs; sigma2; ev = lazy (run inputs ~sigma2 ~targets:None ~want_grad:true ~reuse_v:reused t); reused } in
      t

    let calc inputs ~sigma2 = make inputs sigma2

    (* Model.update_sigma2 (lib/fitc_gp.ml:234-236): K_nm, V and r stay on the device when the problem's last
       evaluation was this model's *)
    let update_sigma2 t sigma2 =
      let reused =
        match !state_owner with
        | Some (p, o) -> Lazy.is_val t.inputs.Inputs.problem && p == Lazy.force t.inputs.Inputs.problem && o == Obj.repr t
        | None -> false
      in
      make ~reused t.inputs sigma2

    let calc_eval t = t
    let calc_log_evidence t = (Lazy.force t.ev).l1
    let calc_log_evidence_sigma2 t = (Lazy.force t.ev).dl_dsigma2
    let prepare_hyper t = ignore (Lazy.force t.ev); t

    let calc_log_evidence_hyper t hyper =
      let ind = t.inputs.Inputs.inducing in
      (Lazy.force t.ev).grad.{1 + S.index_of_hyper ind.Inducing.kernel ~inducing:ind.Inducing.points hyper}

    let ensure_state t =
      match !state_owner with
      | Some (_, o) when o == Obj.repr t -> ()
      | _ -> ignore (run t.inputs ~sigma2:t.sigma2 ~targets:None ~want_grad:false ~reuse_v:false t)

    let calc_co_variance_coeffs t =
      ensure_state t;
      let m = Mat.dim2 t.inputs.Inputs.inducing.Inducing.points in
      let u = Mat.create m m and r = Mat.create m m in
      co_variance_coeffs (Lazy.force t.inputs.Inputs.problem) u r;
      (u, r)

    let get_kernel t = t.inputs.Inputs.inducing.Inducing.kernel
    let get_sigma2 t = t.sigma2
    let get_inputs t = t.inputs
    let get_inducing t = t.inputs.Inputs.inducing
  end

  module Trained = struct
    type t = { model : Model.t; targets : vec; ev : evaluation Lazy.t }
    type hyper_t = t

    let calc (model : Model.t) ~targets =
      if Vec.dim targets <> Mat.dim2 model.Model.inputs.Inputs.points then
        failwith "Trained.calc: Vec.dim targets <> n";
      let rec t =
        {
          model;
          targets;
          ev =
            lazy
              (run model.Model.inputs ~sigma2:model.Model.sigma2 ~targets:(Some targets) ~want_grad:true
                 ~reuse_v:model.Model.reused t);
        }
      in
      t

    let calc_eval t = t
    let calc_mean_coeffs t = (Lazy.force t.ev).coeffs
    let calc_log_evidence t = (Lazy.force t.ev).l
    let calc_log_evidence_sigma2 t = (Lazy.force t.ev).dl_dsigma2
    let prepare_hyper t = ignore (Lazy.force t.ev); t

    let calc_log_evidence_hyper t hyper =
      let ind = t.model.Model.inputs.Inputs.inducing in
      (Lazy.force t.ev).grad.{1 + S.index_of_hyper ind.Inducing.kernel ~inducing:ind.Inducing.points hyper}

    let get_model t = t.model
    let get_targets t = t.targets

    let ensure_state t =
      match !state_owner with
      | Some (_, o) when o == Obj.repr t -> ()
      | _ ->
          ignore
            (run t.model.Model.inputs ~sigma2:t.model.Model.sigma2 ~targets:(Some t.targets) ~want_grad:false
               ~reuse_v:false t)
  end

  (* Means.calc / Variances.calc at new points (lib/fitc_gp.ml:418-425, :498-529) from the trained model's state *)
  module Means = struct
    type t = { points : mat; means : vec }

    let calc (trained : Trained.t) points =
      Trained.ensure_state trained;
      let means = Vec.create (Mat.dim2 points) in
      predict (Lazy.force trained.Trained.model.Model.inputs.Inputs.problem) points false means None;
      { points; means }

    let get t = t.means
  end

  module Variances = struct
    type t = { points : mat; variances : vec; sigma2 : float }

    let calc_model_inputs (trained : Trained.t) points =
      Trained.ensure_state trained;
      let nt = Mat.dim2 points in
      let means = Vec.create nt and variances = Vec.create nt in
      predict (Lazy.force trained.Trained.model.Model.inputs.Inputs.problem) points false means (Some variances);
      { points; variances; sigma2 = trained.Trained.model.Model.sigma2 }

    let get ?(predictive = true) t = if predictive then Vec.add_const t.sigma2 t.variances else t.variances
  end

  (* Stats.calc (lib/fitc_gp.ml:304-374): the residual sums come back from the device, the ratios are formed here *)
  module Stats = struct
    type t = {
      n_samples : int; target_variance : float; sse : float; mse : float; rmse : float; smse : float;
      msll : float; mad : float; maxad : float;
    }

    let calc (trained : Trained.t) =
      Trained.ensure_state trained;
      let sums = Vec.create 4 in
      train_stats (Lazy.force trained.Trained.model.Model.inputs.Inputs.problem) None sums;
      let n = Vec.dim trained.Trained.targets in
      let f_n = float n in
      let y = trained.Trained.targets in
      let mean_y = Vec.sum y /. f_n in
      let target_variance = (sums.{4} /. f_n) -. (mean_y *. mean_y) in
      let sse = sums.{1} in
      let mse = sse /. f_n in
      let l = Trained.calc_log_evidence trained in
      {
        n_samples = n; target_variance; sse; mse; rmse = sqrt mse; smse = mse /. target_variance;
        msll = ((-.l) /. f_n) -. (0.5 *. (log (2. *. Float.pi *. target_variance) +. 1.));
        mad = sums.{2} /. f_n; maxad = sums.{3};
      }
  end

  (* the optimiser callbacks of lib/fitc_gp.ml:1601-1636 over the device evaluation: what Optim.Gsl.train hands to
     Gsl.Multimin.Deriv (parameter vector [log sigma2; hypers], gradient entry 0 scaled by sigma2) *)
  module Optim = struct
    let objective_and_gradient ~inputs ~targets ~sigma2 ~hypers =
      let trained = Trained.calc (Model.calc inputs ~sigma2) ~targets in
      let g = Array.map (fun h -> -.Trained.calc_log_evidence_hyper trained h) hypers in
      (-.Trained.calc_log_evidence trained, -.Trained.calc_log_evidence_sigma2 trained *. sigma2, g)
  end
end

module Se_iso = Make (Iso_spec) (struct let variational = false let precision = f64 let device = 0 end)
module Se_iso_variational = Make (Iso_spec) (struct let variational = true let precision = f64 let device = 0 end)
module Se_fat = Make (Fat_spec) (struct let variational = false let precision = f64 let device = 0 end)
module Se_fat_variational = Make (Fat_spec) (struct let variational = true let precision = f64 let device = 0 end)
module Se_fat_f32 = Make (Fat_spec) (struct let variational = false let precision = f32_bulk let device = 0 end)

(* Signature view.  [Make] keeps the Eval and Deriv faces of a module in one place; the shape of
   Interfaces.Sigs.Deriv (lib/interfaces.ml:848-1154) is obtained by splitting them:

     module Se_iso_sig = struct
       module Eval = struct
         module Spec = Gpr.Cov_se_iso.Eval
         module Inducing = Se_iso.Inducing   module Inputs = Se_iso.Inputs
         module Model = Se_iso.Model         (* calc, update_sigma2, calc_log_evidence, calc_co_variance_coeffs, get_* *)
         module Trained = Se_iso.Trained     (* calc, calc_mean_coeffs, calc_log_evidence, get_* *)
         module Stats = Se_iso.Stats  module Means = Se_iso.Means  module Variances = Se_iso.Variances
       end
       module Deriv = struct
         module Spec = Gpr.Cov_se_iso.Deriv
         module Inducing = Se_iso.Inducing   module Inputs = Se_iso.Inputs     (* calc_eval = identity *)
         module Model = struct
           include Se_iso.Model
           let calc_log_evidence = Se_iso.Model.calc_log_evidence_hyper        (* hyper_t -> Spec.Hyper.t -> float *)
         end
         module Trained = struct
           include Se_iso.Trained
           let calc_log_evidence = Se_iso.Trained.calc_log_evidence_hyper
         end
         module Optim = ...   (* Gsl.train: the reference's driver (lib/fitc_gp.ml:1532-1671) unchanged, its
                                 multim_f / multim_dcommon bodies replaced by Se_iso.Optim.objective_and_gradient *)
       end
     end

   Single-point modules (Input, Mean, Variance), the stored-number predictors (Mean_predictor, Co_variance_predictor),
   Covariances and the samplers follow the same pattern over [predict], [load_predictor], [covariances] and
   [cov_samples]; gpr_amd/fitc_gp.py is their executable counterpart. *)
