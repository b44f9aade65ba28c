// Package ringhip plugs the MI355X ring engine (include/ringhip.h, libringhip.so) into lattigo's ring.Ring through the
// seam the reference already has: NewRingWithCustomNTT(N, moduli, ntt func(*ring.SubRing, int) ring.NumberTheoreticTransformer,
// NthRoot) (ring/ring.go:314-356) and NewSubRingWithCustomNTT (ring/subring.go:74-111).
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (SURVEY F2).  It is the binding a
// maintainer adds on a machine with Go >= 1.21 (runtime.Pinner) and ROCm; see INTEGRATION.md.
package ringhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../matrix-fhe-lattigo_amd/lib -lringhip -Wl,-rpath,${SRCDIR}/../../matrix-fhe-lattigo_amd/lib
#include "ringhip.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"unsafe"

	"github.com/tuneinsight/lattigo/v6/ring"
)

// Transformer implements ring.NumberTheoreticTransformer (ring/ntt.go:17-22) for one SubRing (one modulus).
// The engine handle is created lazily on the first call because the SubRing's NTTTable is filled by
// generateNTTConstants AFTER the factory runs (SURVEY 3.5, ring/ring.go:385-400).
type Transformer struct {
	s      *ring.SubRing
	n      int
	device int
	once   sync.Once
	h      *C.rh_ring
}

// Factory returns the function to pass to ring.NewRingWithCustomNTT.
func Factory(device int) func(*ring.SubRing, int) ring.NumberTheoreticTransformer {
	return func(s *ring.SubRing, n int) ring.NumberTheoreticTransformer {
		return &Transformer{s: s, n: n, device: device}
	}
}

func (t *Transformer) init() {
	t.once.Do(func() {
		s := t.s
		q := C.uint64_t(s.Modulus)
		mred := C.uint64_t(s.MRedConstant)
		bred := [2]C.uint64_t{C.uint64_t(s.BRedConstant[0]), C.uint64_t(s.BRedConstant[1])}
		ninv := C.uint64_t(s.NInv)
		rc := C.rh_ring_create(&t.h, C.int(t.device), C.RH_RING_STANDARD, C.int(t.n), 1, &q, &mred, &bred[0], &ninv,
			(*C.uint64_t)(unsafe.Pointer(&s.RootsForward[0])), (*C.uint64_t)(unsafe.Pointer(&s.RootsBackward[0])), nil)
		if rc != 0 {
			panic(fmt.Sprintf("ringhip: rh_ring_create: %s", C.GoString(C.rh_last_error())))
		}
	})
}

func (t *Transformer) call(f func(*C.rh_ring, C.int, *C.uint64_t, *C.uint64_t) C.int, p1, p2 []uint64) {
	if len(p1) < t.n || len(p2) < t.n { // same contract as ring/ntt.go:212-214
		panic(fmt.Sprintf("cannot NTT: ensure that len(p1)=%d, len(p2)=%d >= N=%d", len(p1), len(p2), t.n))
	}
	t.init()
	// p1/p2 are Go slices of plain uint64: passing &p[0] for the duration of the call is allowed by the cgo rules;
	// the engine copies H2D/D2H inside the call and retains nothing.
	if rc := f(t.h, 0, (*C.uint64_t)(unsafe.Pointer(&p1[0])), (*C.uint64_t)(unsafe.Pointer(&p2[0]))); rc != 0 {
		panic(fmt.Sprintf("ringhip: %s", C.GoString(C.rh_last_error())))
	}
}

func (t *Transformer) Forward(p1, p2 []uint64) {
	t.call(func(h *C.rh_ring, l C.int, a, b *C.uint64_t) C.int { return C.rh_ntt_forward(h, l, a, b) }, p1, p2)
}
func (t *Transformer) ForwardLazy(p1, p2 []uint64) {
	t.call(func(h *C.rh_ring, l C.int, a, b *C.uint64_t) C.int { return C.rh_ntt_forward_lazy(h, l, a, b) }, p1, p2)
}
func (t *Transformer) Backward(p1, p2 []uint64) {
	t.call(func(h *C.rh_ring, l C.int, a, b *C.uint64_t) C.int { return C.rh_ntt_backward(h, l, a, b) }, p1, p2)
}
func (t *Transformer) BackwardLazy(p1, p2 []uint64) {
	t.call(func(h *C.rh_ring, l C.int, a, b *C.uint64_t) C.int { return C.rh_ntt_backward_lazy(h, l, a, b) }, p1, p2)
}

// NewRing is the drop-in for ring.NewRing on power-of-two rings: same Ring type, same methods, NTTs on the GPU.
func NewRing(N int, moduli []uint64, device int) (*ring.Ring, error) {
	return ring.NewRingWithCustomNTT(N, moduli, Factory(device), 2*N)
}

// DeviceRing is the throughput path: a whole ring.Ring (all SubRings at once) mirrored on the device, operating on
// device-resident (poly, limb, coefficient) blocks.  Upload/Download move a ring.Poly ([][]uint64, one Go slice per
// limb: ring/poly.go:13-24) limb by limb.
type DeviceRing struct {
	h *C.rh_ring
	N int
	L int
}

func NewDeviceRing(r *ring.Ring, device int) (*DeviceRing, error) {
	L := r.ModuliChainLength()
	N := r.N()
	mod := make([]C.uint64_t, L)
	mred := make([]C.uint64_t, L)
	bred := make([]C.uint64_t, 2*L)
	ninv := make([]C.uint64_t, L)
	rf := make([]C.uint64_t, L*N)
	rb := make([]C.uint64_t, L*N)
	for i, s := range r.SubRings {
		mod[i], mred[i], ninv[i] = C.uint64_t(s.Modulus), C.uint64_t(s.MRedConstant), C.uint64_t(s.NInv)
		bred[2*i], bred[2*i+1] = C.uint64_t(s.BRedConstant[0]), C.uint64_t(s.BRedConstant[1])
		for j := 0; j < N; j++ {
			rf[i*N+j], rb[i*N+j] = C.uint64_t(s.RootsForward[j]), C.uint64_t(s.RootsBackward[j])
		}
	}
	d := &DeviceRing{N: N, L: L}
	if rc := C.rh_ring_create(&d.h, C.int(device), C.RH_RING_STANDARD, C.int(N), C.int(L), &mod[0], &mred[0], &bred[0],
		&ninv[0], &rf[0], &rb[0], nil); rc != 0 {
		return nil, fmt.Errorf("ringhip: %s", C.GoString(C.rh_last_error()))
	}
	return d, nil
}

// DevPoly is a device block of npoly polynomials with `limbs` limbs.
type DevPoly struct {
	ptr          *C.uint64_t
	npoly, limbs int
}

func (d *DeviceRing) NewPoly(npoly, limbs int) (*DevPoly, error) {
	p := &DevPoly{npoly: npoly, limbs: limbs}
	if rc := C.rh_dev_alloc(d.h, C.size_t(npoly*limbs*d.N), &p.ptr); rc != 0 {
		return nil, fmt.Errorf("ringhip: %s", C.GoString(C.rh_last_error()))
	}
	return p, nil
}

// Upload copies poly k of the block from a ring.Poly, one limb (one Go slice) per call: no Go pointer to Go pointer
// crosses the boundary.
func (d *DeviceRing) Upload(dst *DevPoly, k int, src ring.Poly) {
	for i := 0; i < dst.limbs; i++ {
		off := (k*dst.limbs + i) * d.N
		C.rh_dev_upload(d.h, (*C.uint64_t)(unsafe.Add(unsafe.Pointer(dst.ptr), 8*off)),
			(*C.uint64_t)(unsafe.Pointer(&src.Coeffs[i][0])), C.size_t(d.N))
	}
}
func (d *DeviceRing) Download(dst ring.Poly, src *DevPoly, k int) {
	for i := 0; i < src.limbs; i++ {
		off := (k*src.limbs + i) * d.N
		C.rh_dev_download(d.h, (*C.uint64_t)(unsafe.Pointer(&dst.Coeffs[i][0])),
			(*C.uint64_t)(unsafe.Add(unsafe.Pointer(src.ptr), 8*off)), C.size_t(d.N))
	}
}

// NTT / INTT / MulCoeffsMontgomery ... mirror ring.Ring's methods on device blocks (level = limbs-1).
func (d *DeviceRing) NTT(p1, p2 *DevPoly)  { d.must(C.rh_ring_ntt(d.h, p1.ptr, p2.ptr, C.int(p1.npoly), C.int(p1.limbs-1), 0)) }
func (d *DeviceRing) INTT(p1, p2 *DevPoly) { d.must(C.rh_ring_intt(d.h, p1.ptr, p2.ptr, C.int(p1.npoly), C.int(p1.limbs-1), 0)) }
func (d *DeviceRing) MulCoeffsMontgomery(p1, p2, p3 *DevPoly) {
	d.must(C.rh_ring_vec_op(d.h, C.RH_OP_MUL_MONT, p1.ptr, p2.ptr, p3.ptr, C.int(p3.npoly), C.int(p3.limbs-1), nil, nil))
}
func (d *DeviceRing) MForm(p1, p2 *DevPoly) {
	d.must(C.rh_ring_vec_op(d.h, C.RH_OP_MFORM, p1.ptr, nil, p2.ptr, C.int(p2.npoly), C.int(p2.limbs-1), nil, nil))
}
func (d *DeviceRing) Sync() { d.must(C.rh_ring_sync(d.h)) }
func (d *DeviceRing) must(rc C.int) {
	if rc != 0 {
		panic(fmt.Sprintf("ringhip: %s", C.GoString(C.rh_last_error())))
	}
}

// ---- element-wise ops by opcode (one RH_OP_* per ring/vec_ops.go function) and the callers built on the ring ----------

// VecOp runs any of the 38 element-wise kernels; s0/s1 are per-limb scalars (nil when the op takes none).
func (d *DeviceRing) VecOp(op C.int, p1, p2, p3 *DevPoly, s0, s1 []uint64) {
	var a, b, x, y *C.uint64_t
	if p1 != nil {
		a = p1.ptr
	}
	if p2 != nil {
		b = p2.ptr
	}
	if len(s0) > 0 {
		x = (*C.uint64_t)(unsafe.Pointer(&s0[0]))
	}
	if len(s1) > 0 {
		y = (*C.uint64_t)(unsafe.Pointer(&s1[0]))
	}
	d.must(C.rh_ring_vec_op(d.h, op, a, b, p3.ptr, C.int(p3.npoly), C.int(p3.limbs-1), x, y))
}

// DivRoundByLastModulusManyNTT mirrors ring.Ring.DivRoundByLastModulusManyNTT (ring/scaling.go:130-156).
func (d *DeviceRing) DivRoundByLastModulusManyNTT(nbRescales int, p0, p1 *DevPoly) {
	d.must(C.rh_ring_div_by_last_modulus_many_ntt(d.h, 1 /* round */, C.int(p0.limbs-1), C.int(nbRescales), p0.ptr, p1.ptr, C.int(p1.limbs), C.int(p0.npoly)))
}

// TensorDegree1 runs the degree-1 x degree-1 tensoring of ckks mulRelin (schemes/ckks/evaluator.go:821-834) as one kernel.
func (d *DeviceRing) TensorDegree1(a0, a1, b0, b1, c0, c1, c2 *DevPoly) {
	d.must(C.rh_ring_tensor_degree1(d.h, a0.ptr, a1.ptr, b0.ptr, b1.ptr, c0.ptr, c1.ptr, c2.ptr, C.int(a0.npoly), C.int(a0.limbs-1), 1))
}

// AutomorphismNTT mirrors ring.Ring.AutomorphismNTT (ring/automorphism.go:52-73).
func (d *DeviceRing) AutomorphismNTT(in *DevPoly, galEl uint64, out *DevPoly) {
	d.must(C.rh_ring_automorphism_ntt(d.h, C.int(in.limbs-1), in.ptr, C.uint64_t(galEl), out.ptr, C.int(in.npoly), 0))
}

// KeySwitcher pairs the Q and P device rings (ring.BasisExtender + the gadget product of rlwe.Evaluator).
type KeySwitcher struct {
	be   *C.rh_bext
	q, p *DeviceRing
}

func NewKeySwitcher(q, p *DeviceRing) (*KeySwitcher, error) {
	k := &KeySwitcher{q: q, p: p}
	if rc := C.rh_bext_create(&k.be, q.h, p.h); rc != 0 {
		return nil, fmt.Errorf("ringhip: %s", C.GoString(C.rh_last_error()))
	}
	return k, nil
}

// GadgetProduct mirrors rlwe.Evaluator.GadgetProduct for NTT-domain cx and levelP >= 1
// (core/rlwe/evaluator_gadget_product.go:16-30).  evkQ / evkP hold GadgetCiphertext.Value[i][0][c].Q / .P as
// blocks of beta*2 polys ([digit][component][limb][N]).
func (k *KeySwitcher) GadgetProduct(levelQ, levelP int, cx, evkQ, evkP *DevPoly, beta int, ct0, ct1 *DevPoly) {
	k.q.must(C.rh_bext_gadget_product(k.be, C.int(levelQ), C.int(levelP), cx.ptr, evkQ.ptr, evkP.ptr, C.int(beta), ct0.ptr, ct1.ptr, C.int(cx.npoly)))
}

// GadgetProductThenAdd: ct_c = add_c + GadgetProduct(cx)_c, the ring.Add of Relinearize / mulRelin / Automorphism folded
// into ModDown's epilogue (add0 / add1 may be nil and may be ct0 / ct1 themselves).
func (k *KeySwitcher) GadgetProductThenAdd(levelQ, levelP int, cx, evkQ, evkP *DevPoly, beta int, add0, add1, ct0, ct1 *DevPoly) {
	var a0, a1 *C.uint64_t
	if add0 != nil {
		a0 = add0.ptr
	}
	if add1 != nil {
		a1 = add1.ptr
	}
	k.q.must(C.rh_bext_gadget_product_then_add(k.be, C.int(levelQ), C.int(levelP), cx.ptr, evkQ.ptr, evkP.ptr, C.int(beta), a0, a1,
		ct0.ptr, ct1.ptr, C.int(cx.npoly)))
}

// DecomposeNTT / GadgetProductHoisted mirror the hoisted pair (:431-453, :326-349): one decomposition, many rotations.
func (k *KeySwitcher) DecomposeNTT(levelQ, levelP int, c2 *DevPoly, c2IsNTT bool, decompQ, decompP *DevPoly) {
	isNTT := C.int(0)
	if c2IsNTT {
		isNTT = 1
	}
	k.q.must(C.rh_bext_decompose_ntt(k.be, C.int(levelQ), C.int(levelP), c2.ptr, isNTT, decompQ.ptr, decompP.ptr, C.int(c2.npoly)))
}
func (k *KeySwitcher) GadgetProductHoisted(levelQ, levelP int, decompQ, decompP, evkQ, evkP *DevPoly, beta int, ct0, ct1 *DevPoly) {
	k.q.must(C.rh_bext_gadget_product_hoisted(k.be, C.int(levelQ), C.int(levelP), decompQ.ptr, decompP.ptr, evkQ.ptr, evkP.ptr, C.int(beta),
		ct0.ptr, ct1.ptr, C.int(ct0.npoly)))
}

// ModDownQPtoQNTT mirrors ring.BasisExtender.ModDownQPtoQNTT (ring/basis_extension.go:241-258).
func (k *KeySwitcher) ModDownQPtoQNTT(levelQ, levelP int, p1Q, p1P, p2Q *DevPoly) {
	k.q.must(C.rh_bext_moddown_qp_to_q_ntt(k.be, C.int(levelQ), C.int(levelP), p1Q.ptr, p1P.ptr, p2Q.ptr, C.int(p1Q.npoly)))
}
