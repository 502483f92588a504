// nbco_snap2d.cpp -- snapshot export for the reference's particle viewer.
//
// nbco3 writes headerless fp32 snapshots [pos n x 3 | vel n x 3] (main3.cu:855-858).  The viewer (Graphics/main.cpp:155,181-184)
// plays back files named <folder>/out<20 k>_0.005000.bin, k = 0, 1, 2, ..., each holding DOUBLES: n positions of TWO coordinates
// followed by n velocities of two coordinates (it shows the first half, scaled by 10e4 * 250 to an 8 mm window).  This tool
// projects 3-D snapshots onto two axes and writes that format.
//
//   nbco_snap2d [-axes xy|xz|yz|yx|zx|zy] <in.bin> <out.bin>                     one file
//   nbco_snap2d [-axes ..] -series <in_folder> <steps> <dt> <out_folder>          out<steps k>_<dt>.bin -> out<20 k>_0.005000.bin
//
// Host only (no GPU).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

namespace {

int axis_of(char c) { return c == 'x' ? 0 : (c == 'y' ? 1 : (c == 'z' ? 2 : -1)); }

// 0 ok, 1 input missing, 2 malformed input, 3 cannot write
int convert(const std::string &in, const std::string &out, int a0, int a1)
{
	std::ifstream fin(in, std::ios::in | std::ios::binary | std::ios::ate);
	if (!fin) return 1;
	const std::streamoff len = fin.tellg();
	if (len <= 0 || len % (6 * sizeof(float)) != 0) return 2;
	const size_t n = (size_t)len / (6 * sizeof(float));
	std::vector<float> s(6 * n);
	fin.seekg(0, std::ios::beg);
	fin.read(reinterpret_cast<char *>(s.data()), len);
	std::vector<double> d(4 * n);
	for (size_t i = 0; i < n; ++i)
	{
		d[2 * i] = s[3 * i + a0]; d[2 * i + 1] = s[3 * i + a1];                                 // positions
		d[2 * n + 2 * i] = s[3 * n + 3 * i + a0]; d[2 * n + 2 * i + 1] = s[3 * n + 3 * i + a1];   // velocities
	}
	std::ofstream fout(out, std::ios::out | std::ios::binary);
	if (!fout) return 3;
	fout.write(reinterpret_cast<const char *>(d.data()), (std::streamsize)(d.size() * sizeof(double)));
	return fout ? 0 : 3;
}

const char *kUsage =
    "Usage: nbco_snap2d [-axes xy] <in.bin> <out.bin>\n"
    "       nbco_snap2d [-axes xy] -series <in_folder> <steps> <dt> <out_folder>\n"
    "  Projects nbco3 snapshots (fp32 xyz positions then velocities) onto two axes and writes the particle viewer's format\n"
    "  (doubles: n x 2 positions, n x 2 velocities); -series renames out<steps k>_<dt>.bin to out<20 k>_0.005000.bin.\n";

} // namespace

int main(int argc, char **argv)
{
	int a0 = 0, a1 = 1;
	std::vector<std::string> args;
	bool series = false;
	for (int i = 1; i < argc; ++i)
	{
		const std::string a(argv[i]);
		if (a == "-h" || a == "-help") { std::cout << kUsage; return 0; }
		if (a == "-axes")
		{
			if (i + 1 >= argc || std::strlen(argv[i + 1]) != 2 || axis_of(argv[i + 1][0]) < 0 || axis_of(argv[i + 1][1]) < 0 || argv[i + 1][0] == argv[i + 1][1])
			{
				std::cerr << "Error: invalid argument to '-axes'\n";
				return -1;
			}
			a0 = axis_of(argv[i + 1][0]); a1 = axis_of(argv[i + 1][1]);
			++i;
		}
		else if (a == "-series") series = true;
		else args.push_back(a);
	}
	if (!series)
	{
		if (args.size() != 2) { std::cerr << kUsage; return -1; }
		const int rc = convert(args[0], args[1], a0, a1);
		if (rc == 1) std::cerr << "Error: cannot read from input location." << std::endl;
		if (rc == 2) std::cerr << "Error: the input is not a [pos | vel] fp32 snapshot." << std::endl;
		if (rc == 3) std::cerr << "Error: cannot write on output location." << std::endl;
		return rc ? -1 : 0;
	}
	if (args.size() != 4) { std::cerr << kUsage; return -1; }
	const int steps = std::atoi(args[1].c_str());
	if (steps <= 0) { std::cerr << "Error: invalid <steps>\n"; return -1; }
	const std::string dt = std::to_string(std::atof(args[2].c_str()));   // as nbco3 names its files (main3.cu:858)
	int frames = 0;
	for (;; ++frames)
	{
		const std::string in = args[0] + "/out" + std::to_string((long long)frames * steps) + '_' + dt + ".bin";
		const std::string out = args[3] + "/out" + std::to_string(frames * 20) + '_' + std::to_string(0.005) + ".bin";   // Graphics/main.cpp:155
		const int rc = convert(in, out, a0, a1);
		if (rc == 1) break;
		if (rc == 2) { std::cerr << "Error: " << in << " is not a [pos | vel] fp32 snapshot." << std::endl; return -1; }
		if (rc == 3) { std::cerr << "Error: cannot write on output location. Check that \"" << args[3] << "\" folder exists. Create it if not." << std::endl; return -1; }
	}
	std::cout << frames << " frame(s) written" << std::endl;
	return frames > 0 ? 0 : -1;
}
