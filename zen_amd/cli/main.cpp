// zen -- command line tool of the MI355X HPSS engine: `zen offline`, `zen fakert`, help, version.
// Same sub-commands, flags, defaults and console lines as the reference's zen/main.cu:12-92,
// zen/offline.h and zen/fakert.h (clipp grammar re-done by hand; libnyquist replaced by cli/wav.h).
// Differences: the `compute:` / timing lines name this backend, and `--cpu` is refused -- this tool is
// the GPU drop-in; the CPU restatement lives under oracle/ as test infrastructure only.
#include <algorithm>
#include <array>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dirent.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <map>
#include <string>
#include <vector>

#include <sched.h>
#include <sys/wait.h>
#include <unistd.h>

#include <libzen/hps.h>
#include <libzen/io.h>
#include <libzen/zen.h>

#include "wav.h"

namespace {

struct OfflineParams { // zen/offline.h:19-32
	std::string infile = "", outfile_prefix = "";
	bool do_hps = false, cpu = false, nocopybord = false, use_sse = false, soft_mask = false, only_percussive = false;
	std::size_t hop_h = 4096, hop_p = 256;
	float beta_h = 2.0, beta_p = 2.0;
};

struct FakeRtParams { // zen/fakert.h:39-49
	std::string infile = "", outfile = "";
	bool do_hps = false, cpu = false, nocopybord = false, use_sse = false, soft_mask = false;
	std::size_t hop = 256;
	float beta = 2.0;
	int resident_ms = 0; // --resident <ms>: the MI355X resident-kernel extension (HPRRealtime::use_resident_kernel); 0: off
};

void usage(std::ostream& os)
{
	os << "usage:\n\n"
	      "  zen offline -i <infile> [--hps [<hop-h> [<beta-h> [<hop-p> [<beta-p>]]]]] [-o <outfile_prefix>]\n"
	      "      [--sse] [--only-percussive] [--soft-mask] [--nocopybord]\n"
	      "  zen fakert -i <infile> [--hps [<hop> [<beta>]]] [-o <outfile>] [--sse] [--soft-mask] [--nocopybord]\n"
	      "      [--resident <idle-ms>]\n"
	      "  zen batch -i <indir> -o <outdir> [--hps [<hop-h> [<beta-h> [<hop-p> [<beta-p>]]]]] [--sse] [--soft-mask]\n"
	      "      [--gpus <n>]\n"
	      "      (MI355X extension: every .wav of <indir>, equal-length clips separated together in one batch;\n"
	      "       --gpus n deals the files round-robin to n processes, one per GPU, no exchange between them)\n"
	      "  zen help|-h|--help\n"
	      "  zen version|-v|--version\n";
}

bool is_number(const char* s)
{
	char* end = nullptr;
	std::strtod(s, &end);
	return end != s && *end == '\0';
}

std::vector<float> load_mono(const std::string& path, zen::wav::AudioData& fd)
{
	zen::wav::load(fd, path);
	std::cout << "Audio file info:" << std::endl;
	std::cout << "\tsample rate: " << fd.sampleRate << std::endl;
	std::cout << "\tlen samples: " << fd.samples.size() << std::endl;
	std::cout << "\tframe size: " << fd.frameSize << std::endl;
	std::cout << "\tseconds: " << fd.lengthSeconds << std::endl;
	std::cout << "\tchannels: " << fd.channelCount << std::endl;
	if (fd.channelCount == 2) { // offline.h:106-113
		std::vector<float> mono(fd.samples.size() / 2);
		zen::wav::stereo_to_mono(fd.samples.data(), mono.data(), fd.samples.size());
		return mono;
	}
	return fd.samples;
}

void peak_normalise(std::vector<float>& x, std::size_t n) // offline.h:182-191, fakert.h:260-269
{
	auto limits = std::minmax_element(x.begin(), x.end());
	const float real_max = std::max(-1 * (*limits.first), *limits.second);
	for (std::size_t j = 0; j < n; ++j)
		x[j] /= real_max;
}

int run_offline(const OfflineParams& p)
{
	std::cout << "Running zen-offline with the following params:"
	          << "\n\tinfile: " << p.infile << "\n\toutfile_prefix: " << p.outfile_prefix
	          << "\n\tonly_percussive: " << p.only_percussive;
	if (p.do_hps) {
		std::cout << "\n\tdo hps: yes"
		          << "\n\t\tharmonic hop: " << p.hop_h << "\n\t\tharmonic beta: " << p.beta_h
		          << "\n\t\tpercussive hop: " << p.hop_p << "\n\t\tpercussive beta: " << p.beta_p;
		std::cout << (p.soft_mask ? "\n\t\tmask: soft/Wiener" : "\n\t\tmask: hard/binary");
		std::cout << (p.use_sse ? "\n\t\tfilter: sse" : "\n\t\tfilter: median");
	}
	else {
		std::cout << "\n\tdo hps: no";
	}
	std::cout << "\n\tcompute: gpu (hip/gfx950)" << std::endl;

	zen::wav::AudioData fd;
	std::vector<float> audio = load_mono(p.infile, fd);
	std::array<std::vector<float>, 3> all_out;
	const std::size_t n_audio = audio.size();
	if (p.do_hps) {
		std::cout << "Processing input signal of size " << audio.size()
		          << " with HPR-I separation using harmonic params: " << p.hop_h << "," << p.beta_h
		          << ", percussive params: " << p.hop_p << "," << p.beta_p << std::endl;
		zen::hps::HPRIOffline<zen::Backend::GPU> hpss((float)fd.sampleRate, p.hop_h, p.hop_p, p.beta_h, p.beta_p,
		                                              p.nocopybord);
		if (p.use_sse)
			hpss.use_sse_filter();
		if (p.soft_mask)
			hpss.use_soft_mask();
		auto t1 = std::chrono::high_resolution_clock::now();
		all_out = hpss.process(std::move(audio)); // (offline.h:144 passes an lvalue: a copy of the clip the tool never reads again)
		auto t2 = std::chrono::high_resolution_clock::now();
		auto dur = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
		std::cout << "GPU/HIP/gfx950: 2-pass HPR-I-Offline took " << dur << " ms" << std::endl;
	}
	else {
		all_out = {audio, audio, audio};
	}
	static const char* suffix[3] = {"_harm.wav", "_perc.wav", "_residual.wav"};
	if (p.outfile_prefix != "") {
		for (int i = 0; i < 3; ++i) {
			if (p.only_percussive && i != 1)
				continue;
			peak_normalise(all_out[i], n_audio);
			zen::wav::encode_pcm16_mono(all_out[i], fd.sampleRate, p.outfile_prefix + suffix[i]);
		}
	}
	return 0;
}

// `zen batch`: the many-clip offline case (BASELINE configs[3]) from the command line.  Clips of equal
// sample rate and length go through zen_hip_hpri_process_device together (n_clips streams in lock step);
// outputs are written as <outdir>/<stem>_harm.wav and <stem>_perc.wav with the same peak normalisation
// and PCM16 encoding as `zen offline`.
#define ZEN_CK(call)                                                        \
	do {                                                                    \
		if ((call) != ZEN_HIP_OK)                                           \
			throw std::runtime_error(std::string(#call ": ") + zen_hip_last_error()); \
	} while (0)

// Rank / world of this process inside `zen batch --gpus N` (set by run_batch_multi for its children).
struct BatchRank {
	int rank = 0, world = 1, report_fd = -1;
};

BatchRank batch_rank_from_env()
{
	BatchRank b;
	if (const char* r = std::getenv("ZEN_BATCH_RANK"))
		b.rank = std::atoi(r);
	if (const char* w = std::getenv("ZEN_BATCH_WORLD"))
		b.world = std::max(1, std::atoi(w));
	if (const char* f = std::getenv("ZEN_BATCH_REPORT_FD"))
		b.report_fd = std::atoi(f);
	return b;
}

// CPUs next to GPU `index` (the order HIP enumerates: the GPU nodes of the KFD topology), from sysfs alone -- no HIP call:
// node N of /sys/class/kfd/kfd/topology/nodes is a GPU when its simd_count > 0, its drm_render_minor M names
// /sys/class/drm/renderD<M>/device, whose local_cpulist is the answer (zen_amd/dist.py gpu_numa_cpus: the same walk).
// Empty when the box does not say, when ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES remap the devices, or ZEN_NO_NUMA_BIND is set.
std::vector<int> gpu_numa_cpus(int index)
{
	std::vector<int> cpus;
	if (std::getenv("ZEN_NO_NUMA_BIND") || std::getenv("ROCR_VISIBLE_DEVICES") || std::getenv("HIP_VISIBLE_DEVICES"))
		return cpus;
	const char* env_root = std::getenv("ZEN_SYSFS_ROOT"); // (tests)
	const std::string sysfs = env_root ? env_root : "/sys";
	const std::string root = sysfs + "/class/kfd/kfd/topology/nodes";
	std::vector<int> nodes;
	if (DIR* d = opendir(root.c_str())) {
		while (dirent* e = readdir(d))
			if (std::isdigit((unsigned char)e->d_name[0]))
				nodes.push_back(std::atoi(e->d_name));
		closedir(d);
	}
	std::sort(nodes.begin(), nodes.end());
	int seen = 0, minor = -1;
	for (int nd : nodes) {
		std::ifstream f(root + "/" + std::to_string(nd) + "/properties");
		std::string k;
		long long v, simd = 0, rm = -1;
		while (f >> k >> v) {
			if (k == "simd_count")
				simd = v;
			if (k == "drm_render_minor")
				rm = v;
		}
		if (simd > 0 && seen++ == index)
			minor = (int)rm;
	}
	if (minor < 0)
		return cpus;
	const std::string dev = sysfs + "/class/drm/renderD" + std::to_string(minor) + "/device/";
	int node = -1;
	std::ifstream(dev + "numa_node") >> node;
	if (node < 0)
		return cpus;
	std::ifstream lf(dev + "local_cpulist");
	std::string list;
	std::getline(lf, list);
	std::stringstream ss(list);
	std::string part;
	while (std::getline(ss, part, ',')) {
		int lo = 0, hi = 0;
		if (std::sscanf(part.c_str(), "%d-%d", &lo, &hi) == 2)
			for (int c = lo; c <= hi; ++c)
				cpus.push_back(c);
		else if (std::sscanf(part.c_str(), "%d", &lo) == 1)
			cpus.push_back(lo);
	}
	return cpus;
}

// Pin the calling process to the CPUs next to its GPU, before its first GPU call: the offline path moves 12 bytes per sample
// between pageable host vectors and the device (zen/offline.h:141-147); a child whose buffers sit on the other socket pays the
// inter-socket link for every one of them.
void bind_to_gpu_numa(int index)
{
	const std::vector<int> cpus = gpu_numa_cpus(index);
	if (cpus.empty())
		return;
	cpu_set_t set;
	CPU_ZERO(&set);
	for (int c : cpus)
		if (c >= 0 && c < CPU_SETSIZE)
			CPU_SET(c, &set);
	(void)sched_setaffinity(0, sizeof(set), &set);
}

// `zen batch --gpus N` (BASELINE configs[3]): clips are independent units, so the parent only
// starts N copies of itself -- one per GPU, before anything in this process has touched a GPU -- and adds
// up what they report through a pipe.  Child r separates files r, r+N, r+2N, ... of the sorted directory
// listing on device r.  No collective: every child reads its own inputs and writes its own outputs.
int run_batch_multi(int gpus, char* argv[])
{
	struct Child {
		pid_t pid;
		int fd;
	};
	std::vector<Child> kids;
	const auto t1 = std::chrono::high_resolution_clock::now();
	for (int r = 0; r < gpus; ++r) {
		int fds[2];
		if (pipe(fds) != 0)
			throw std::runtime_error("pipe failed");
		const pid_t pid = fork();
		if (pid < 0)
			throw std::runtime_error("fork failed");
		if (pid == 0) {
			close(fds[0]);
			setenv("ZEN_BATCH_RANK", std::to_string(r).c_str(), 1);
			setenv("ZEN_BATCH_WORLD", std::to_string(gpus).c_str(), 1);
			setenv("ZEN_BATCH_REPORT_FD", std::to_string(fds[1]).c_str(), 1);
			bind_to_gpu_numa(r); // (inherited across the exec: the child's threads and first-touched pages stay next to GPU r)
			execv("/proc/self/exe", argv);
			_exit(127);
		}
		close(fds[1]);
		kids.push_back({pid, fds[0]});
	}
	double audio_s = 0, gpu_ms_max = 0;
	long files = 0;
	int rc = 0;
	for (auto& k : kids) {
		std::string text;
		char buf[256];
		ssize_t got;
		while ((got = read(k.fd, buf, sizeof(buf))) > 0)
			text.append(buf, (size_t)got);
		close(k.fd);
		int status = 0;
		waitpid(k.pid, &status, 0);
		if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
			rc = WIFEXITED(status) ? WEXITSTATUS(status) : 1;
		double a = 0, ms = 0;
		long f = 0;
		if (std::sscanf(text.c_str(), "%lf %lf %ld", &a, &ms, &f) == 3) {
			audio_s += a;
			gpu_ms_max = std::max(gpu_ms_max, ms);
			files += f;
		}
	}
	const auto t2 = std::chrono::high_resolution_clock::now();
	const long wall = (long)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
	std::cout << "zen batch: " << gpus << " GPUs: " << files << " files, " << audio_s << " s of audio separated; "
	          << gpu_ms_max << " ms of GPU time on the busiest GPU, " << wall << " ms wall incl. file I/O" << std::endl;
	return rc;
}

int run_batch(const OfflineParams& p, const std::string& outdir, const BatchRank& br)
{
	const std::string tag = br.world > 1 ? "[gpu " + std::to_string(br.rank) + "] " : "";
	std::vector<std::string> files;
	if (DIR* d = opendir(p.infile.c_str())) {
		while (dirent* e = readdir(d)) {
			const std::string name = e->d_name;
			if (name.size() > 4 && name.substr(name.size() - 4) == ".wav")
				files.push_back(name);
		}
		closedir(d);
	}
	else {
		throw std::runtime_error("cannot open directory " + p.infile);
	}
	std::sort(files.begin(), files.end());
	if (br.world > 1) { // this rank's share: files rank, rank + world, ... (same deal as zen_amd/dist.py shard_units)
		std::vector<std::string> mine;
		for (std::size_t i = (std::size_t)br.rank; i < files.size(); i += (std::size_t)br.world)
			mine.push_back(files[i]);
		files.swap(mine);
	}
	struct Clip {
		std::string stem;
		std::vector<float> mono;
	};
	std::map<std::pair<int, std::size_t>, std::vector<Clip>> groups; // (sample rate, length) -> clips
	for (const auto& f : files) {
		zen::wav::AudioData fd;
		zen::wav::load(fd, p.infile + "/" + f);
		Clip c;
		c.stem = f.substr(0, f.size() - 4);
		if (fd.channelCount == 2) {
			c.mono.resize(fd.samples.size() / 2);
			zen::wav::stereo_to_mono(fd.samples.data(), c.mono.data(), fd.samples.size());
		}
		else {
			c.mono = fd.samples;
		}
		if (!c.mono.empty())
			groups[{fd.sampleRate, c.mono.size()}].push_back(std::move(c));
	}
	std::cout << tag << "zen batch: " << files.size() << " wav files in " << groups.size() << " (rate, length) groups"
	          << std::endl;
	const std::size_t max_batch = 64;
	double total_audio_s = 0;
	long total_ms = 0;
	for (auto& g : groups) {
		const int fs = g.first.first;
		const std::size_t n = g.first.second;
		auto& clips = g.second;
		for (std::size_t c0 = 0; c0 < clips.size(); c0 += max_batch) {
			const std::size_t C = std::min(max_batch, clips.size() - c0);
			zen_hip_hpri_t eng = nullptr;
			int rc = zen_hip_hpri_create((float)fs, p.hop_h, p.hop_p, p.beta_h, p.beta_p, p.nocopybord ? 1 : 0, C, &eng);
			if (rc == ZEN_HIP_E_HOPS_NOT_DIVISIBLE)
				throw zen::ZgException("hop_h and hop_p should be evenly divisible");
			ZEN_CK(rc);
			if (p.use_sse)
				ZEN_CK(zen_hip_hpri_use_sse_filter(eng));
			if (p.soft_mask)
				ZEN_CK(zen_hip_hpri_use_soft_mask(eng));
			void *d_in = nullptr, *d_h = nullptr, *d_p = nullptr;
			ZEN_CK(zen_hip_malloc(&d_in, C * n * sizeof(float)));
			ZEN_CK(zen_hip_malloc(&d_h, C * n * sizeof(float)));
			ZEN_CK(zen_hip_malloc(&d_p, C * n * sizeof(float)));
			for (std::size_t c = 0; c < C; ++c)
				ZEN_CK(zen_hip_memcpy_h2d((float*)d_in + c * n, clips[c0 + c].mono.data(), n * sizeof(float)));
			auto t1 = std::chrono::high_resolution_clock::now();
			ZEN_CK(zen_hip_hpri_process_device(eng, (const float*)d_in, n, n, (float*)d_h, (float*)d_p, nullptr, n));
			ZEN_CK(zen_hip_synchronize(nullptr));
			auto t2 = std::chrono::high_resolution_clock::now();
			const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
			total_ms += ms;
			total_audio_s += (double)C * n / fs;
			std::cout << tag << "GPU/HIP/gfx950: 2-pass HPR-I-Offline of " << C << " clips x " << n << " samples took " << ms
			          << " ms" << std::endl;
			std::vector<float> out(n);
			for (std::size_t c = 0; c < C; ++c) {
				for (int which = 0; which < 2; ++which) {
					ZEN_CK(zen_hip_memcpy_d2h(out.data(), (float*)(which ? d_p : d_h) + c * n, n * sizeof(float)));
					peak_normalise(out, n);
					zen::wav::encode_pcm16_mono(out, fs, outdir + "/" + clips[c0 + c].stem + (which ? "_perc.wav" : "_harm.wav"));
				}
			}
			zen_hip_free(d_in);
			zen_hip_free(d_h);
			zen_hip_free(d_p);
			zen_hip_hpri_destroy(eng);
		}
	}
	std::cout << tag << "zen batch: " << total_audio_s << " s of audio separated in " << total_ms << " ms of GPU time"
	          << std::endl;
	if (br.report_fd >= 0) { // to run_batch_multi
		const std::string line = std::to_string(total_audio_s) + " " + std::to_string(total_ms) + " "
		                         + std::to_string(files.size()) + "\n";
		if (write(br.report_fd, line.data(), line.size()) < 0)
			std::cerr << tag << "zen batch: report pipe closed" << std::endl;
		close(br.report_fd);
	}
	return 0;
}

// zen/fakert.h:15-34 -- emits only whole chunks and stops before the last one
std::vector<std::pair<std::size_t, std::size_t>> get_chunk_limits(const std::vector<float>& container, std::size_t k)
{
	std::vector<std::pair<std::size_t, std::size_t>> ret;
	const std::size_t size = container.size();
	std::size_t i = 0;
	if (size > k)
		for (; i < size - k; i += k)
			ret.push_back({i, i + k});
	if (i % k)
		ret.push_back({i, i + (i % k)});
	return ret;
}

int run_fakert(const FakeRtParams& p)
{
	std::cout << "Running zen-fakert with the following params:"
	          << "\n\tinfile: " << p.infile << "\n\toutfile: " << p.outfile;
	if (p.do_hps) {
		std::cout << "\n\tdo hps: yes"
		          << "\n\t\thop: " << p.hop << "\n\t\tbeta: " << p.beta;
		std::cout << (p.soft_mask ? "\n\t\tmask: soft/Wiener" : "\n\t\tmask: hard/binary");
		std::cout << (p.use_sse ? "\n\t\tfilter: sse" : "\n\t\tfilter: median");
	}
	else {
		std::cout << "\n\tdo hps: no";
	}
	std::cout << "\n\tcompute: gpu (hip/gfx950)" << std::endl;

	zen::wav::AudioData fd;
	std::vector<float> audio = load_mono(p.infile, fd);
	std::vector<float> percussive_out = audio; // fakert.h:132: the unprocessed tail stays raw input
	const auto chunk_limits = get_chunk_limits(audio, p.hop);
	std::cout << "Slicing buffer size " << audio.size() << " into " << chunk_limits.size() << " chunks of size "
	          << p.hop << std::endl;

	std::size_t n = 0;
	const float delta_t = 1000 * (float)p.hop / fd.sampleRate;
	zen::hps::HPRRealtime<zen::Backend::GPU> hpss((float)fd.sampleRate, p.hop, p.beta, zen::hps::OUTPUT_PERCUSSIVE,
	                                              p.nocopybord);
	zen::io::IOGPU io(p.hop);
	if (p.use_sse)
		hpss.use_sse_filter();
	if (p.soft_mask)
		hpss.use_soft_mask();
	hpss.warmup(io);
	if (p.resident_ms > 0) // after the warm-up (its reset would send the kernel home anyway)
		hpss.use_resident_kernel(p.resident_ms);

	float iters = 0.0F;
	long time_tot = 0;
	for (const auto& chunk : chunk_limits) {
		auto t1 = std::chrono::high_resolution_clock::now();
		if (p.do_hps) {
			std::copy(audio.begin() + chunk.first, audio.begin() + chunk.second, io.host_in);
			hpss.process_next_hop(io.device_in);
			hpss.copy_percussive(io.device_out);
			std::copy(io.host_out, io.host_out + p.hop, percussive_out.begin() + n);
		}
		else {
			std::copy(audio.begin() + chunk.first, audio.begin() + chunk.second, percussive_out.begin() + n);
		}
		auto t2 = std::chrono::high_resolution_clock::now();
		time_tot += std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
		n += p.hop;
		iters += 1.0F;
	}
	std::cout << "PRealtime GPU:  Δn = " << p.hop << ", Δt(ms) = " << delta_t
	          << ", average processing duration(us) = " << (float)time_tot / iters << std::endl;

	if (p.outfile != "") {
		peak_normalise(percussive_out, audio.size());
		zen::wav::encode_pcm16_mono(percussive_out, fd.sampleRate, p.outfile);
	}
	return 0;
}

} // namespace

int main(int argc, char* argv[])
{
	std::vector<std::string> a(argv + 1, argv + argc);
	if (a.empty()) {
		usage(std::cerr);
		return 0;
	}
	const std::string cmd = a[0];
	if (cmd == "help" || cmd == "-h" || cmd == "--help") {
		usage(std::cout);
		return 0;
	}
	if (cmd == "version" || cmd == "-v" || cmd == "--version") {
		std::cout << "version 1.0\n";
		return 0;
	}
	if (cmd != "offline" && cmd != "fakert" && cmd != "batch") {
		usage(std::cerr);
		return 0;
	}
	OfflineParams op;
	FakeRtParams fp;
	bool cpu = false, have_input = false;
	int gpus = 1;
	for (std::size_t i = 1; i < a.size(); ++i) {
		const std::string& s = a[i];
		auto next_is_number = [&]() { return i + 1 < a.size() && is_number(a[i + 1].c_str()); };
		if ((s == "-i" || s == "--input") && i + 1 < a.size()) {
			op.infile = fp.infile = a[++i];
			have_input = true;
		}
		else if (s == "--hps") {
			op.do_hps = fp.do_hps = true;
			if (cmd != "fakert") {
				if (next_is_number()) op.hop_h = std::strtoul(a[++i].c_str(), nullptr, 10);
				if (next_is_number()) op.beta_h = std::strtof(a[++i].c_str(), nullptr);
				if (next_is_number()) op.hop_p = std::strtoul(a[++i].c_str(), nullptr, 10);
				if (next_is_number()) op.beta_p = std::strtof(a[++i].c_str(), nullptr);
			}
			else {
				if (next_is_number()) fp.hop = std::strtoul(a[++i].c_str(), nullptr, 10);
				if (next_is_number()) fp.beta = std::strtof(a[++i].c_str(), nullptr);
			}
		}
		else if ((s == "-o" || (cmd != "fakert" && s == "--out-prefix") || (cmd == "fakert" && s == "--output"))
		         && i + 1 < a.size()) {
			op.outfile_prefix = fp.outfile = a[++i];
		}
		else if (s == "--gpus" && cmd == "batch" && next_is_number()) gpus = std::max(1, std::atoi(a[++i].c_str()));
		else if (s == "--cpu") cpu = true;
		else if (s == "--sse") op.use_sse = fp.use_sse = true;
		else if (s == "--soft-mask") op.soft_mask = fp.soft_mask = true;
		else if (s == "--nocopybord") op.nocopybord = fp.nocopybord = true;
		else if (s == "--resident" && cmd == "fakert" && next_is_number()) fp.resident_ms = std::atoi(a[++i].c_str());
		else if (s == "--only-percussive" && cmd == "offline") op.only_percussive = true;
		else {
			usage(std::cerr);
			return 0;
		}
	}
	if (!have_input) {
		usage(std::cerr);
		return 0;
	}
	if (cpu) {
		std::cerr << "zen: --cpu is not available in the MI355X drop-in (GPU backend only)" << std::endl;
		return 2;
	}
	try {
		const BatchRank br = batch_rank_from_env();
		if (cmd == "batch" && op.outfile_prefix == "") {
			usage(std::cerr);
			return 0;
		}
		if (cmd == "batch" && gpus > 1 && br.world == 1)
			return run_batch_multi(gpus, argv); // before zen_hip_init: the parent never touches a GPU
		int device = cmd == "batch" ? br.rank : 0;
		if (cmd == "batch" && br.world > 1 && std::getenv("ZEN_ALLOW_GPU_SHARING")) {
			int n_dev = 0; // testing on a box with fewer GPUs than ranks: ranks share devices
			if (zen_hip_device_count(&n_dev) == ZEN_HIP_OK && n_dev > 0)
				device = br.rank % n_dev;
		}
		if (zen_hip_init(device) != ZEN_HIP_OK) {
			std::cerr << "zen: " << zen_hip_last_error() << std::endl;
			return 1;
		}
		if (cmd == "batch")
			return run_batch(op, op.outfile_prefix, br);
		return cmd == "offline" ? run_offline(op) : run_fakert(fp);
	}
	catch (const std::exception& e) {
		std::cerr << "zen: " << e.what() << std::endl;
		return 1;
	}
}
