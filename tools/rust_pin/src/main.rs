//! reef_rust_pin: one JSON document on stdout, in the layout of tests/golden/, holding what the crates Reef links
//! (fil_pasta_curves 0.5.2 `repr-c`, pasta-msm, neptune 8.1, sga001/Nova) really do for the inputs this repository's oracle
//! and HIP kernels are tested on.  `cargo run --release > ../../tests/golden/rust_pin.json`, then
//! `python -m pytest tests/test_pin_from_rust.py` (CPU: the oracle; `-m gpu`: the HIP path through the C ABI).
//!
//! It could NOT be compiled where it was written (no Rust toolchain, no network): API spellings marked `[R]` are recalled
//! and may need a one-line fix against the crate versions your Cargo.lock resolves.  Every section is independent: comment
//! out the one that does not compile, the test skips what is absent.
//!
//! Sections and the [R] facts of include/reef_msm.h they pin:
//!   layout            size_of / raw bytes of Fp, Fq, EpAffine, Ep: 4 x u64 little-endian Montgomery limbs (R = 2^256),
//!                     identity encodings -- (0, 0) affine, z = 0 projective
//!   msm               the seeded cases of tests/golden/pasta_msm_golden.json through pasta_msm::pallas / vesta (the Rust
//!                     wrapper passes is_mont = true) AND through the naive definition, compressed
//!   compress          GroupEncoding::to_bytes of 2G, 3G, 5G, 100G, (q-1)G: little-endian x, y parity in bit 255
//!   commitment_gens   CommitmentGens::new(b"ck", 8) of the fork, each generator compressed (from_label: SHAKE256 + hash_to_curve)
//!   poseidon          neptune's Sponge<Fq, U4> constants (Strength::Standard): round constants, MDS, the domain tags of the
//!                     IO patterns Reef uses, the four `new_parent` query shapes of src/backend/merkle_tree.rs:82-114, the
//!                     Merkle root of the document of its own test `make_mt` (merkle_tree.rs:209-257)
//!   linear_mle        three rounds of src/backend/r1cs_helper.rs:441-506 on a fixed 8-entry table: coefficients AND the
//!                     challenges the sponge squeezes (IO pattern of r1cs.rs:2260-2284 with k = 2 absorbed first)
use ff::{Field, PrimeField};
use generic_array::typenum::U4;
use group::{prime::PrimeCurveAffine, Curve, Group, GroupEncoding};
use neptune::sponge::api::{IOPattern, SpongeAPI, SpongeOp};
use neptune::sponge::vanilla::{Mode, Sponge, SpongeTrait};
use neptune::Strength;
use pasta_curves::{pallas, vesta};
use serde_json::{json, Value};
use sha2::{Digest, Sha256};

// ---- the generators of oracle/pasta_oracle.py (SplitMix64, uniform_scalar, witness_like, ap_bases) -----------------------
struct SplitMix64(u64);
impl SplitMix64 {
    fn next(&mut self) -> u64 {
        self.0 = self.0.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        z ^ (z >> 31)
    }
}
/// little-endian 32 bytes -> field element, reducing once if the 255-bit value is >= the modulus (oracle: uniform_scalar)
fn fe_from_le_reduce_once<F: PrimeField<Repr = [u8; 32]>>(bytes: [u8; 32]) -> F {
    match Option::<F>::from(F::from_repr(bytes)) {
        Some(f) => f,
        None => {
            // v - modulus, computed on the bytes: modulus = -1 + 1 as field elements is not available without big ints, so
            // subtract through u64 limbs against the modulus taken from F::MODULUS (a "0x.." big-endian hex string) [R]
            let m = hex::decode(F::MODULUS.trim_start_matches("0x")).expect("MODULUS hex");
            let mut mle = [0u8; 32];
            for (i, b) in m.iter().rev().enumerate() {
                mle[i] = *b;
            }
            let mut out = [0u8; 32];
            let mut borrow = 0i16;
            for i in 0..32 {
                let d = bytes[i] as i16 - mle[i] as i16 - borrow;
                out[i] = d.rem_euclid(256) as u8;
                borrow = if d < 0 { 1 } else { 0 };
            }
            Option::<F>::from(F::from_repr(out)).expect("reduced once")
        }
    }
}
fn uniform_scalar<F: PrimeField<Repr = [u8; 32]>>(rng: &mut SplitMix64) -> F {
    let mut b = [0u8; 32];
    for w in 0..4 {
        b[8 * w..8 * w + 8].copy_from_slice(&rng.next().to_le_bytes());
    }
    b[31] &= 0x7f;
    fe_from_le_reduce_once::<F>(b)
}
fn witness_like<F: PrimeField<Repr = [u8; 32]>>(rng: &mut SplitMix64) -> F {
    let w: Vec<u64> = (0..4).map(|_| rng.next()).collect();
    let sel = (w[3] >> 56) % 10;
    if sel < 7 {
        F::from(w[0] & 1)
    } else if sel < 9 {
        F::from(w[0] & 0xFFFF)
    } else {
        let mut b = [0u8; 32];
        for i in 0..4 {
            b[8 * i..8 * i + 8].copy_from_slice(&w[i].to_le_bytes());
        }
        b[31] &= 0x7f;
        fe_from_le_reduce_once::<F>(b)
    }
}
fn hexs(b: impl AsRef<[u8]>) -> String {
    hex::encode(b.as_ref())
}
/// the raw in-memory bytes of a `repr-c` value: what crosses the C ABI of libreef_msm.so
fn raw<T>(t: &T) -> String {
    let p = t as *const T as *const u8;
    hexs(unsafe { std::slice::from_raw_parts(p, std::mem::size_of::<T>()) })
}

macro_rules! curve_sections {
    ($name:expr, $m:ident, $msm:path) => {{
        type P = $m::Point;
        type A = $m::Affine;
        type S = $m::Scalar;
        type B = $m::Base;
        let g = P::generator();
        let layout = json!({
            "curve": $name,
            "size_of_base": std::mem::size_of::<B>(), "size_of_scalar": std::mem::size_of::<S>(),
            "size_of_affine": std::mem::size_of::<A>(), "size_of_point": std::mem::size_of::<P>(),
            "align_of_affine": std::mem::align_of::<A>(),
            "base_one_raw": raw(&B::one()),             // = R mod p as 4 x u64 LE if elements are kept in Montgomery form
            "scalar_one_raw": raw(&S::one()),
            "scalar_two_repr": hexs(S::from(2u64).to_repr()),   // canonical little-endian 02 00 ..
            "affine_identity_raw": raw(&A::identity()),  // expected: 64 zero bytes, i.e. (0, 0)
            "point_identity_raw": raw(&P::identity()),   // expected: z = 0
            "affine_generator_raw": raw(&g.to_affine()), // x = -1, y = 2 in Montgomery form
            "point_generator_raw": raw(&g),              // 96 bytes: x, y, z
        });
        // --- compress
        let mut comp = vec![];
        let qm1 = -S::one();
        for (label, k) in [("0x2", S::from(2u64)), ("0x3", S::from(3u64)), ("0x5", S::from(5u64)), ("0x64", S::from(100u64)), ("order-1", qm1)] {
            comp.push(json!({"curve": $name, "k": label, "compressed": hexs((g * k).to_affine().to_bytes())}));
        }
        // --- the seeded MSM cases of oracle/gen_golden.py::seeded_cases
        let mut msm = vec![];
        for n in [1usize, 2, 3, 127, 128, 129, 1000, 4096] {
            for kind in [0u64, 1] {
                if kind == 1 && ![128usize, 1000, 4096].contains(&n) {
                    continue;
                }
                let seed = 0x5EEFu64 + n as u64 + 1000 * kind;
                let (k0, d) = (7 + n as u64, 3u64);
                let mut rng = SplitMix64(seed);
                let sc: Vec<S> = (0..n).map(|_| if kind == 0 { uniform_scalar::<S>(&mut rng) } else { witness_like::<S>(&mut rng) }).collect();
                let mut bases_p = Vec::with_capacity(n);
                let (mut cur, step) = (g * S::from(k0), g * S::from(d));
                for _ in 0..n {
                    bases_p.push(cur);
                    cur += step;
                }
                let mut bases = vec![A::identity(); n];
                P::batch_normalize(&bases_p, &mut bases);
                // the hash the Python side recomputes over ITS inputs: canonical affine (x || y little-endian) then canonical scalars
                let mut h = Sha256::new();
                for b in &bases {
                    let c = b.coordinates().unwrap();   // [R] pasta_curves::arithmetic::CurveAffine::coordinates
                    h.update(c.x().to_repr());
                    h.update(c.y().to_repr());
                }
                for s in &sc {
                    h.update(s.to_repr());
                }
                let naive = bases_p.iter().zip(sc.iter()).fold(P::identity(), |acc, (b, s)| acc + *b * *s);
                let fast = $msm(&bases, &sc);           // the extern "C" mult_pippenger_* seam (is_mont = true inside the wrapper)
                msm.push(json!({"curve": $name, "n": n, "seed": seed, "kind": kind, "k0": k0, "d": d,
                                "input_sha256": hexs(h.finalize()),
                                "expect_compressed": hexs(naive.to_affine().to_bytes()),
                                "pasta_msm_compressed": hexs(fast.to_affine().to_bytes()),
                                "pasta_msm_point_raw": raw(&fast)}));
            }
        }
        (layout, comp, msm)
    }};
}

fn fq_hex(f: &pallas::Scalar) -> String {
    format!("0x{}", hexs(f.to_repr().iter().rev().cloned().collect::<Vec<u8>>()))
}

fn main() {
    let (lay_p, comp_p, msm_p) = curve_sections!("pallas", pallas, pasta_msm::pallas);
    let (lay_v, comp_v, msm_v) = curve_sections!("vesta", vesta, pasta_msm::vesta);

    // ---- commitment generators of the fork: CommitmentGens::new(label, n) -> from_label (SHAKE256 + hash_to_curve) [R]
    let commitment_gens: Value = {
        use nova_snark::provider::pedersen::CommitmentGens;
        use nova_snark::traits::{commitment::CommitmentEngineTrait, Group as NovaGroup};
        type G1 = pallas::Point;
        let n = 8usize;
        let gens = CommitmentGens::<G1>::new(b"ck", n);
        let zero = <G1 as NovaGroup>::Scalar::zero();
        let mut out = vec![];
        for i in 0..n {
            let mut v = vec![zero; n];
            v[i] = <G1 as NovaGroup>::Scalar::one();
            // Reef's call shape (src/backend/commitment.rs:350): CE::commit(&gens, &values, &blind); blind = 0 isolates G_i
            let c = <G1 as NovaGroup>::CE::commit(&gens, &v, &zero);
            out.push(hexs(c.compress().to_bytes()));   // [R] CompressedCommitment bytes = GroupEncoding::to_bytes
        }
        json!({"label": "ck", "n": n, "curve": "pallas", "generators_compressed": out})
    };

    // ---- neptune: constants, tags, the four new_parent shapes, make_mt's root
    type F = pallas::Scalar;
    let pc = Sponge::<F, U4>::api_constants(Strength::Standard);
    let tag = |pattern: Vec<SpongeOp>| -> String {
        // [R] the capacity element of SpongeAPI::start: IOPattern::value(domain_separator = 0) as a field element
        let v: u128 = IOPattern(pattern).value(0);
        fq_hex(&F::from_u128(v))
    };
    let hash = |query: &[F]| -> F {
        let mut sponge = Sponge::new_with_constants(&pc, Mode::Simplex);
        let acc = &mut ();
        sponge.start(IOPattern(vec![SpongeOp::Absorb(query.len() as u32), SpongeOp::Squeeze(1)]), None, acc);
        SpongeAPI::absorb(&mut sponge, query.len() as u32, query, acc);
        let out = SpongeAPI::squeeze(&mut sponge, 1, acc);
        sponge.finish(acc).unwrap();
        out[0]
    };
    let f = |x: u64| F::from(x);
    let shapes: Vec<Vec<F>> = vec![vec![f(0), f(7), f(1), f(9)], vec![f(4), f(3), F::zero(), F::zero()], vec![f(11), f(12)], vec![f(13), F::zero()]];
    let sponge_outputs: Vec<Value> =
        shapes.iter().map(|q| json!({"query": q.iter().map(fq_hex).collect::<Vec<_>>(), "out": fq_hex(&hash(q))})).collect();
    // make_mt (merkle_tree.rs:209-257): doc = [2, 3, 4, 5, 6, 7, 8]; leaves H4(2i, doc[2i], 2i+1, doc[2i+1]), odd tail (i, c, 0, 0); parents H2
    let doc: Vec<u64> = (2..=8).collect();
    let mut level: Vec<F> = doc
        .chunks(2)
        .enumerate()
        .map(|(i, c)| if c.len() == 2 { hash(&[f(2 * i as u64), f(c[0]), f(2 * i as u64 + 1), f(c[1])]) } else { hash(&[f(2 * i as u64), f(c[0]), F::zero(), F::zero()]) })
        .collect();
    let mut levels = vec![level.iter().map(fq_hex).collect::<Vec<_>>()];
    while level.len() > 1 {
        level = level.chunks(2).map(|c| if c.len() == 2 { hash(&[c[0], c[1]]) } else { hash(&[c[0], F::zero()]) }).collect();
        levels.push(level.iter().map(fq_hex).collect());
    }
    let poseidon = json!({
        "field": "pallas scalar field (Fq)", "arity": 4, "width": 5,
        "full_rounds": pc.full_rounds, "partial_rounds": pc.partial_rounds,          // [R] public fields of PoseidonConstants
        "round_constants": pc.round_constants.as_ref().expect("round constants").iter().map(fq_hex).collect::<Vec<_>>(),   // round-major, width per round
        "mds": pc.mds_matrices.m.iter().map(|row| row.iter().map(fq_hex).collect::<Vec<_>>()).collect::<Vec<_>>(),        // [R] m[i][j]; new[j] = sum_i state[i] * m[i][j]
        "tag_leaf": tag(vec![SpongeOp::Absorb(4), SpongeOp::Squeeze(1)]),
        "tag_node": tag(vec![SpongeOp::Absorb(2), SpongeOp::Squeeze(1)]),
        "sponge_outputs": sponge_outputs,
        "make_mt": {"doc": doc, "levels": levels},
    });

    // ---- linear_mle_product: 3 rounds on a fixed table, challenges from the sponge (r1cs.rs:2260-2284: Absorb(k), Squeeze(1), then (Absorb(3), Squeeze(1)) x ell)
    let linear_mle: Value = {
        let ell = 3usize;
        let mut t: Vec<F> = (0..8u64).map(|i| f(3 * i * i + 5 * i + 1)).collect();
        let mut e: Vec<F> = (0..8u64).map(|i| f(1000 + 17 * i * i * i)).collect();
        let first: Vec<F> = vec![f(424242), f(31337)];
        let mut pattern = vec![SpongeOp::Absorb(first.len() as u32), SpongeOp::Squeeze(1)];
        for _ in 0..ell {
            pattern.extend([SpongeOp::Absorb(3), SpongeOp::Squeeze(1)]);
        }
        let tag_hex = tag(pattern.clone());
        let mut sponge = Sponge::new_with_constants(&pc, Mode::Simplex);
        let acc = &mut ();
        sponge.start(IOPattern(pattern), None, acc);
        SpongeAPI::absorb(&mut sponge, first.len() as u32, &first, acc);
        let claim_r = SpongeAPI::squeeze(&mut sponge, 1, acc)[0];
        let table_t: Vec<String> = t.iter().map(fq_hex).collect();
        let table_eq: Vec<String> = e.iter().map(fq_hex).collect();
        let mut rounds = vec![];
        for i in 1..=ell {
            let pow = 1usize << (ell - i);
            let (mut xsq, mut x, mut con) = (F::zero(), F::zero(), F::zero());
            for b in 0..pow {
                let (ts, es) = (t[b + pow] - t[b], e[b + pow] - e[b]);
                xsq += ts * es;
                x += es * t[b] + ts * e[b];
                con += t[b] * e[b];
            }
            SpongeAPI::absorb(&mut sponge, 3, &[con, x, xsq], acc);       // the order of r1cs_helper.rs:478-482
            let r = SpongeAPI::squeeze(&mut sponge, 1, acc)[0];
            for b in 0..pow {
                t[b] = t[b] * (F::one() - r) + t[b + pow] * r;
                e[b] = e[b] * (F::one() - r) + e[b + pow] * r;
            }
            rounds.push(json!({"xsq": fq_hex(&xsq), "x": fq_hex(&x), "con": fq_hex(&con), "r": fq_hex(&r)}));
        }
        sponge.finish(acc).unwrap();
        json!({"ell": ell, "table_t": table_t, "table_eq": table_eq, "first_absorb": first.iter().map(fq_hex).collect::<Vec<_>>(),
               "first_squeeze": fq_hex(&claim_r), "tag": tag_hex, "rounds": rounds, "t_final": fq_hex(&t[0]), "eq_final": fq_hex(&e[0])})
    };

    // ---- Reef's own frontend and cost model (feature reef-frontend): the automaton SAFA::new builds for the regexes oracle/gen_replay_shapes.py
    //      takes from the reference's scripts, and costs.rs evaluated on it -- what oracle/safa_shape.py and oracle/costs_oracle.py restate
    #[cfg(feature = "reef-frontend")]
    let reef_frontend: Value = {
        use reef::backend::costs::{full_round_cost_model, opt_cost_model_select};
        use reef::frontend::regex::re;
        use reef::frontend::safa::SAFA;
        let ascii: String = (0u32..128).filter_map(std::char::from_u32).collect();   // src/config.rs:229-233
        let brca1_a = "ATGGGCTACAGAAACCGTGCCAAAAGACTTCTACAGAGTGAACCCGAAAATCCTTCCTTG";   // tests/scripts/dna.sh:6
        let brca1_b = ["ATGCTGAAACTTCTCAACCAGAAGAAAGGGCCTTCACAGTGTCCTTTATGTAAGAATGATATAACCAAAAG",
                       "AGCCTACAAGAAAGTACGAGATTTAGTCAACTTGTTGAAGAGCTATTGAAAATCATTTGTGCTTTTCAGCTTGACACAGGTTTGGAGT",
                       "ATGCAAACAGCTATAATTTTGCAAAAAAGGAAAATAACTCTCCTGAACATCTAAAAGATGAAGTTTCTATCATCCAAAGTATGGGCTACAGAAACCGTGCCAAAAGACTTCTACAGAGTGAACCCGAAAATCCTTCCTTG"];   // dna.sh:7
        let doc16: usize = 1 << 24;
        let cases: Vec<(String, String, usize, bool, usize)> = vec![      // (regex, alphabet, document bytes, hybrid, batch; 0 = chosen by the model)
            (".*b".into(), ascii.clone(), 9, false, 0),
            (".*password.*".into(), ascii.clone(), 1 << 20, false, 0),
            (format!("^.{{{}}}{}", doc16 - 10000 + 8129, brca1_a), "ACGT".into(), doc16, true, 32),
            (format!("^.{{{}}}{}", doc16 - 10000 + 5784, brca1_b.join(".*")), "ACGT".into(), doc16, true, 32),
            ("^baa$".into(), "ab".into(), 3, false, 2),          // the automata of safa.rs's own tests (safa.rs:574-610)
            ("baa".into(), "ab".into(), 8, false, 2),
        ];
        let mut out = vec![];
        for (rx, ab, doc_bytes, hybrid, batch) in cases {
            let r = re::simpl(re::new(&rx));
            let safa = SAFA::new(&ab, &r);
            let udoc_len = (doc_bytes + 2).next_power_of_two();          // framework.rs:997-1008
            let max_offsets = safa.max_skip_offset().max(1) + 2;          // r1cs.rs:108-110
            let hybrid_len = if hybrid { Some(2 * udoc_len.max(safa.num_edges().next_power_of_two())) } else { None };   // r1cs.rs:481-487 (the table is far smaller than the document here)
            // the solution lengths NFA::new would hand the model are not reachable without building the whole converter; the cost at a FIXED batch is
            let b = if batch == 0 { 2 } else { batch };
            let cost = full_round_cost_model(&safa, b, udoc_len, hybrid, hybrid_len, false, max_offsets, 1, 1);
            out.push(json!({"regex": if rx.len() > 120 { format!("{}...{}", &rx[..60], &rx[rx.len() - 40..]) } else { rx.clone() }, "regex_len": rx.len(),
                            "alphabet_size": ab.chars().count(), "doc_bytes": doc_bytes, "hybrid": hybrid, "batch": b,
                            "num_states": safa.num_states(), "num_edges": safa.num_edges(), "max_skip_offset": safa.max_skip_offset(),
                            "full_round_cost_model": cost}));
        }
        let _ = opt_cost_model_select;   // (its inputs need NFA::new's path lengths; see tests/test_safa_shape.py for how the oracle derives them)
        json!(out)
    };
    #[cfg(not(feature = "reef-frontend"))]
    let reef_frontend: Value = Value::Null;

    let doc = json!({
        "generated_by": "tools/rust_pin (cargo run --release); see tools/rust_pin/README.md",
        "crates": {"fil_pasta_curves": "0.5.2 (repr-c)", "pasta-msm": env!("CARGO_PKG_VERSION"), "neptune": "8.1.0", "nova-snark": "git sga001/Nova (state the rev)"},
        "layout": [lay_p, lay_v],
        "compress": comp_p.into_iter().chain(comp_v).collect::<Vec<_>>(),
        "msm": msm_p.into_iter().chain(msm_v).collect::<Vec<_>>(),
        "commitment_gens": commitment_gens,
        "poseidon": poseidon,
        "linear_mle": linear_mle,
        "reef_frontend": reef_frontend,
    });
    println!("{}", serde_json::to_string_pretty(&doc).unwrap());
}
